"""Generates tests/golden/losses_*.npz by importing the REFERENCE criterion (/root/reference/models/losses.py:
torch + scipy only) in this container and running it on seeded inputs.  The reference cannot travel; the
vectors (inputs, losses, matched indices, gradients of the total loss w.r.t. every prediction) do.

    python tests/golden/make_losses_golden.py

Cases: matcher weights of main_utils.py:243 (1, 0, 2, soft token) and the class defaults (1, 5, 2), with
and without the contrastive term, padded utterances, scenes with few / many / zero valid targets.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/models/losses.py"


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_losses", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_inputs(seed, B, Q, G, C, L, D, K, N, layers, n_valid, topk):
    """Seeded end_points-like dict (CPU tensors) with every key compute_hungarian_loss reads."""
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, generator=g)
    uni = lambda *s: torch.rand(*s, generator=g)
    prefixes = ["proposal_", "last_"] + [f"{i}head_" for i in range(layers - 1)]
    ep = {}
    mask = torch.zeros(B, G)
    for b in range(B):
        mask[b, :n_valid[b]] = 1
    ep["box_label_mask"] = mask
    ep["center_label"] = uni(B, G, 3) * 4 - 2
    ep["size_gts"] = uni(B, G, 3) * 1.5 + 0.2
    ep["sem_cls_label"] = torch.randint(0, C - 1, (B, G), generator=g)
    # positive maps: a normalised span of 1-3 tokens per target
    pm = torch.zeros(B, G, C)
    lens = torch.randint(max(4, L - 4), L + 1, (B,), generator=g)      # real tokens incl. <s> and </s>
    for b in range(B):
        for t in range(G):
            start = int(torch.randint(1, int(lens[b]) - 2, (1,), generator=g))
            width = int(torch.randint(1, 4, (1,), generator=g))
            end = min(start + width, int(lens[b]) - 1)
            pm[b, t, start:end] = 1.0 / (end - start)
    ep["positive_map"] = pm
    att = (torch.arange(L)[None, :] < lens[:, None]).long()
    ep["tokenized"] = {"attention_mask": att}
    tok = rnd(B, L, D)
    ep["proj_tokens"] = tok / tok.norm(dim=-1, keepdim=True)
    for p in prefixes:
        ep[f"{p}center"] = uni(B, Q, 3) * 4 - 2
        ep[f"{p}pred_size"] = uni(B, Q, 3) * 1.5 + 0.1
        ep[f"{p}sem_cls_scores"] = rnd(B, Q, C) * 2
        q = rnd(B, Q, D)
        ep[f"{p}proj_queries"] = q / q.norm(dim=-1, keepdim=True)
    # seeds: every valid target owns >= topk seeds (no top-k ties among the 100-filled entries)
    ep["seed_xyz"] = uni(B, K, 3) * 4 - 2
    ep["seed_inds"] = torch.stack([torch.randperm(N, generator=g)[:K] for _ in range(B)]).int()
    pil = torch.full((B, N), -1, dtype=torch.long)
    for b in range(B):
        owners = torch.full((K,), -1, dtype=torch.long)
        nv = int(n_valid[b])
        if nv:
            owners[: nv * topk] = torch.arange(nv).repeat_interleave(topk)
            extra = torch.randint(-1, nv, (K - nv * topk,), generator=g)
            owners[nv * topk:] = extra
        pil[b, ep["seed_inds"][b].long()] = owners
    ep["point_instance_label"] = pil
    ep["seeds_obj_cls_logits"] = rnd(B, 1, K)
    return ep, prefixes


def run_case(ref, name, weights, use_contrastive, **shape):
    ep, prefixes = make_inputs(**shape)
    layers, topk = shape["layers"], shape["topk"]
    if not use_contrastive:
        for k in [k for k in ep if k.endswith("proj_queries")] + ["proj_tokens"]:
            del ep[k]
    leaves = {}
    for k, v in ep.items():
        if torch.is_tensor(v) and v.dtype == torch.float32 and any(
                k.endswith(s) for s in ("center", "pred_size", "sem_cls_scores", "proj_queries")) or k in (
                "proj_tokens", "seeds_obj_cls_logits"):
            ep[k] = v.clone().requires_grad_(True)
            leaves[k] = ep[k]
    matcher = ref.HungarianMatcher(*weights, True)
    losses = ["boxes", "labels"] + (["contrastive_align"] if use_contrastive else [])
    crit = ref.SetCriterion(matcher=matcher, losses=losses, eos_coef=0.1, temperature=0.07)
    import torch.distributed as dist
    if not dist.is_initialized():     # SetCriterion.forward calls dist.get_world_size() unconditionally
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29581", rank=0, world_size=1)
    inputs = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in ep.items()}
    loss, out = ref.compute_hungarian_loss(dict(ep), layers, crit, topk)
    loss.backward()
    save = {}
    for k, v in inputs.items():
        if k == "tokenized":
            save["in_attention_mask"] = v["attention_mask"].numpy()
        else:
            save["in_" + k] = v.numpy()
    for k in ("loss", "loss_ce", "loss_bbox", "loss_giou", "query_points_generation_loss",
              "loss_constrastive_align"):
        save["out_" + k] = np.asarray(float(out[k]), dtype=np.float64)
    for p in prefixes:
        for key in ("loss_ce", "loss_bbox", "loss_giou", "loss_contrastive_align"):
            if f"{p}_{key}" in out:
                save[f"out_{p}_{key}"] = np.asarray(float(out[f"{p}_{key}"]), dtype=np.float64)
    # the assignment of every prefix in the dense form: match[p, b, slot] = query (-1 = not a target)
    B, G = inputs["box_label_mask"].shape
    match = -np.ones((len(prefixes), B, G), dtype=np.int32)
    tgt = [{"labels": inputs["sem_cls_label"][b, inputs["box_label_mask"][b].bool()],
            "boxes": torch.cat([inputs["center_label"], inputs["size_gts"]], -1)[b, inputs["box_label_mask"][b].bool()],
            "positive_map": inputs["positive_map"][b, inputs["box_label_mask"][b].bool()]} for b in range(B)]
    for i, p in enumerate(prefixes):
        o = {"pred_logits": inputs[f"{p}sem_cls_scores"],
             "pred_boxes": torch.cat([inputs[f"{p}center"], inputs[f"{p}pred_size"]], -1)}
        for b, (qi, ti) in enumerate(matcher(o, tgt)):
            slots = torch.nonzero(inputs["box_label_mask"][b]).flatten()
            match[i, b, slots[ti].numpy()] = qi.numpy()
    save["out_match"] = match
    for k, v in leaves.items():
        save["grad_" + k] = v.grad.numpy() if v.grad is not None else np.zeros(tuple(v.shape), np.float32)
    save["meta_layers"] = np.asarray(layers)
    save["meta_topk"] = np.asarray(topk)
    save["meta_weights"] = np.asarray(weights, dtype=np.float64)
    save["meta_contrastive"] = np.asarray(int(use_contrastive))
    path = os.path.join(HERE, f"losses_{name}.npz")
    np.savez_compressed(path, **save)
    print(name, "loss", float(loss), "->", path, os.path.getsize(path) // 1024, "KiB")


BENCH_SHAPE = dict(B=8, Q=256, G=132, C=256, L=80, D=64, K=1024, N=50000, layers=6, topk=4, seed=5,
                   n_valid=[1, 16, 5, 132, 66, 9, 3, 12])


def run_bench_case(ref):
    """The criterion at the bench's size (8 scenes x 256 queries x 7 prefixes, 132 target slots holding 1 .. 132 targets,
    256 token classes, 1024 seeds of 50 000 points) with the training run's matcher weights: losses, the assignment of every
    prefix and the gradients w.r.t. every prediction.  The inputs are NOT stored (47 MB): the test regenerates them with
    make_inputs(**BENCH_SHAPE) (torch's CPU generator is deterministic); the gradients are stored as every 8th row."""
    weights = (1, 0, 2)
    ep, prefixes = make_inputs(**BENCH_SHAPE)
    leaves = {}
    for k, v in ep.items():
        if torch.is_tensor(v) and v.dtype == torch.float32 and any(
                k.endswith(s) for s in ("center", "pred_size", "sem_cls_scores", "proj_queries")) or k in (
                "proj_tokens", "seeds_obj_cls_logits"):
            ep[k] = v.clone().requires_grad_(True)
            leaves[k] = ep[k]
    matcher = ref.HungarianMatcher(*weights, True)
    crit = ref.SetCriterion(matcher=matcher, losses=["boxes", "labels", "contrastive_align"], eos_coef=0.1, temperature=0.07)
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29581", rank=0, world_size=1)
    inputs = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in ep.items()}
    loss, out = ref.compute_hungarian_loss(dict(ep), BENCH_SHAPE["layers"], crit, BENCH_SHAPE["topk"])
    loss.backward()
    save = {}
    for k in ("loss", "loss_ce", "loss_bbox", "loss_giou", "query_points_generation_loss", "loss_constrastive_align"):
        save["out_" + k] = np.asarray(float(out[k]), dtype=np.float64)
    for p in prefixes:
        for key in ("loss_ce", "loss_bbox", "loss_giou", "loss_contrastive_align"):
            if f"{p}_{key}" in out:
                save[f"out_{p}_{key}"] = np.asarray(float(out[f"{p}_{key}"]), dtype=np.float64)
    B, G = inputs["box_label_mask"].shape
    match = -np.ones((len(prefixes), B, G), dtype=np.int32)
    tgt = [{"labels": inputs["sem_cls_label"][b, inputs["box_label_mask"][b].bool()],
            "boxes": torch.cat([inputs["center_label"], inputs["size_gts"]], -1)[b, inputs["box_label_mask"][b].bool()],
            "positive_map": inputs["positive_map"][b, inputs["box_label_mask"][b].bool()]} for b in range(B)]
    for i, p in enumerate(prefixes):
        o = {"pred_logits": inputs[f"{p}sem_cls_scores"],
             "pred_boxes": torch.cat([inputs[f"{p}center"], inputs[f"{p}pred_size"]], -1)}
        for b, (qi, ti) in enumerate(matcher(o, tgt)):
            slots = torch.nonzero(inputs["box_label_mask"][b]).flatten()
            match[i, b, slots[ti].numpy()] = qi.numpy()
    save["out_match"] = match
    for k, v in leaves.items():
        gk = v.grad if v.grad is not None else torch.zeros_like(v)
        save["grad8_" + k] = (gk[:, ::8] if gk.dim() == 3 and gk.shape[1] >= 64 else gk).numpy()
        save["gradsum_" + k] = np.asarray(float(gk.double().abs().sum()))
    path = os.path.join(HERE, "criterion_bench_size.npz")
    np.savez_compressed(path, **save)
    print("bench size: loss", float(loss), "->", path, os.path.getsize(path) // 1024, "KiB")


def main():
    if not os.path.exists(REF):
        sys.exit("the reference is not mounted here")
    ref = load_reference()
    base = dict(B=3, Q=24, G=7, C=32, L=12, D=8, K=64, N=400, layers=3, topk=3)
    run_case(ref, "train_weights", (1, 0, 2), True, seed=1, n_valid=[3, 7, 1], **base)
    run_case(ref, "default_weights", (1, 5, 2), True, seed=2, n_valid=[5, 2, 6], **base)
    run_case(ref, "no_contrastive", (1, 0, 2), False, seed=3, n_valid=[4, 4, 2], **base)
    run_case(ref, "empty_scene", (1, 5, 2), True, seed=4, n_valid=[0, 3, 5], **base)
    run_bench_case(ref)


if __name__ == "__main__":
    if sys.argv[1:] == ["bench_size"]:
        run_bench_case(load_reference())
    else:
        main()
