"""criterion pieces: kernel launches and summed device time (torch.profiler), B=8 north-star shapes"""
import torch, numpy as np
from torch.profiler import profile, ProfilerActivity
from butd_detr_amd import losses as L
from butd_detr_amd.train_step import synthetic_ground_truth
torch.manual_seed(0)
dev = "cuda"
P, B, Q, C, Lt, D, G, K, N = 7, 8, 256, 256, 80, 64, 132, 1024, 50000
rng = np.random.default_rng(0)
pc = rng.uniform(-3, 3, (B, N, 3)).astype(np.float32)
gt = {k: torch.from_numpy(v).to(dev) for k, v in synthetic_ground_truth(pc, rng).items()}
logits = torch.randn(P, B, Q, C, device=dev, requires_grad=True)
boxes = torch.cat([torch.rand(P, B, Q, 3, device=dev) * 6 - 3, torch.rand(P, B, Q, 3, device=dev) + 0.2], -1).requires_grad_(True)
pq = torch.nn.functional.normalize(torch.randn(P, B, Q, D, device=dev), dim=-1).requires_grad_(True)
tok = torch.nn.functional.normalize(torch.randn(B, Lt, D, device=dev), dim=-1).requires_grad_(True)
att = torch.ones(B, Lt, dtype=torch.long, device=dev)
crit = L.SetCriterion(L.HungarianMatcher(1, 0, 2, True), ["boxes", "labels", "contrastive_align"])
tgt = {"boxes": torch.cat([gt["center_label"], gt["size_gts"]], -1), "positive_map": gt["positive_map"],
       "labels": gt["sem_cls_label"], "valid": gt["box_label_mask"] > 0}
out = {"pred_logits": logits, "pred_boxes": boxes, "proj_queries": pq, "proj_tokens": tok, "tokenized": {"attention_mask": att}}
nb = crit.num_boxes(tgt["valid"])
match = crit.matcher.match_dense(logits, boxes, tgt["boxes"], tgt["positive_map"], tgt["valid"])
ep = {"box_label_mask": gt["box_label_mask"], "seed_inds": torch.randint(0, N, (B, K), device=dev).int(),
      "seed_xyz": torch.rand(B, K, 3, device=dev) * 6 - 3, "seeds_obj_cls_logits": torch.randn(B, 1, K, device=dev, requires_grad=True),
      "center_label": gt["center_label"], "size_gts": gt["size_gts"], "point_instance_label": gt["point_instance_label"]}
def prof(name, fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as p:
        fn(); torch.cuda.synchronize()
    ev = [e for e in p.events() if e.device_type.name == "CUDA"]
    print(f"{name:28s} kernels {len(ev):4d}  device time {sum(e.device_time for e in ev)/1e3:7.3f} ms")
    if name.startswith("contrastive") or name.startswith("cost") or name.startswith("labels"):
        for e in sorted(ev, key=lambda e: -e.device_time)[:8]:
            print(f"      {e.device_time:7.1f} us  {e.name[:90]}")
def bw(x): x.sum().backward()
with torch.no_grad():
    prof("cost (torch)", lambda: crit.matcher.cost(logits, boxes, tgt["boxes"], tgt["positive_map"]))
    c = crit.matcher.cost(logits, boxes, tgt["boxes"], tgt["positive_map"])
    prof("lsap", lambda: L.hungarian_match(c, tgt["valid"]))
prof("labels fwd+bwd", lambda: bw(crit.loss_labels_st(out, tgt, match, nb)["loss_ce"]))
prof("boxes fwd+bwd", lambda: bw(sum(crit.loss_boxes(out, tgt, match, nb).values())))
prof("contrastive fwd+bwd", lambda: bw(crit.loss_contrastive_align(out, tgt, match, nb)["loss_contrastive_align"]))
prof("objectness fwd+bwd", lambda: L.compute_points_obj_cls_loss_hard_topk(ep, 4).backward())
