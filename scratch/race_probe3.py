"""round 3 race probe: per-step loss / clip coefficient logged ON THE DEVICE (no host read-back between steps).
LR0=1: learning rates 0 and the dropout counter pinned per batch, so every visit of a batch must reproduce the
same numbers -- any deviation is a corrupted step, not a chaotic trajectory."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = sys.argv[:1]
import torch, bench
from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, synthetic_batch
from butd_detr_amd import fused_attention as fa
args = bench.parse()
dev = torch.device("cuda", 0)
E = os.environ.get
if E("SPLIT") == "1":
    import torch.distributed as dist
    os.environ["BUTD_FORCE_COLLECTIVE"] = "1"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group(E("BACKEND", "nccl"), init_method="env://", rank=0, world_size=1)
model, _ = bench.build_model(args, dev)
crit = bench.make_criterion(args)
model.text_encoder.eval()
for m in model.text_projector.modules():
    if isinstance(m, torch.nn.Dropout): m.p = 0.0
lr0 = E("LR0") == "1"
opt = FlatAdamW(model, lr=0.0, lr_backbone=0.0, text_encoder_lr=0.0, weight_decay=0.0) if lr0 else FlatAdamW(model)
step = GraphedTrainStep(model, opt, criterion=crit, prefetch_sampling=E("PS", "1") == "1",
                        prefetch_text=E("PT", "1") == "1", overlap_exchange=E("OVERLAP", "0") == "1")
nb = int(E("NBATCH", "3"))
batches = [synthetic_batch(args.batch, dev, seed=1184 + 50 * i, n_points=args.points, tokens=args.tokens) for i in range(nb)]
n = int(E("STEPS", "60"))
log = torch.zeros(n, 3, device=dev)
ctr = fa.rng_counter(dev)
for it in range(n):
    inp, tgt = batches[it % nb]
    if lr0:
        ctr.fill_(1000 + it % nb)
    loss = step(inp, tgt, next_inputs=batches[(it + 1) % nb][0])
    log[it, 0].copy_(loss)
    log[it, 1].copy_(opt.grad_scale[0])
    log[it, 2].copy_(opt.flat_g[::97].abs().sum())
torch.cuda.synchronize()
log = log.cpu().double()
tag = E("TAG", "probe")
if lr0:
    bad = []
    for j in range(nb):
        seq = log[j::nb]
        ref = seq[1] if len(seq) > 1 else seq[0]          # (visit 0 of batch 0 is the capture call)
        dev_ = ((seq - ref).abs() / ref.abs().clamp_min(1e-12)).max(1).values
        bad += [(j + nb * i, float(d)) for i, d in enumerate(dev_) if d > 1e-4]
    print(tag, "LR0 deviating steps (step, rel):", bad[:40], "count", len(bad), "of", n)
    print(tag, "first rows", log[:6].tolist())
else:
    print(tag, "loss seq", [round(float(v), 4) for v in log[:, 0]])
os.makedirs("gpurun_out", exist_ok=True)
torch.save(log, f"gpurun_out/probe_{tag}.pt")
