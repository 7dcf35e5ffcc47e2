"""Region timing of one training step (GPU events + host wall) for B=8."""
import sys, time, warnings, argparse
sys.path.insert(0, '.')
import torch
from bench import build_model
from butd_detr_amd.train_step import make_optimizer, synthetic_batch, surrogate_loss
args = argparse.Namespace(backend=sys.argv[1] if len(sys.argv) > 1 else "torch", queries=256, points=50000, tokens=80)
dev = torch.device("cuda", 0)
model, backend = build_model(args, dev)
opt = make_optimizer(model)
inputs, targets = synthetic_batch(8, dev)
regions = []
class T:
    def __init__(s, name): s.name = name
    def __enter__(s):
        s.e0 = torch.cuda.Event(enable_timing=True); s.e1 = torch.cuda.Event(enable_timing=True)
        s.t0 = time.perf_counter(); s.e0.record(); return s
    def __exit__(s, *a):
        s.e1.record(); s.t1 = time.perf_counter(); regions.append(s)
def hook(mod, name):
    orig = mod.forward
    def f(*a, **k):
        with T(name):
            return orig(*a, **k)
    mod.forward = f
m = model
for n in ("sa1", "sa2", "sa3", "sa4", "fp1", "fp2"):
    hook(getattr(m.backbone_net, n), "fwd." + n)
hook(m.text_encoder, "fwd.roberta"); hook(m.cross_encoder, "fwd.encoder")
for i, l in enumerate(m.decoder): hook(l, "fwd.decoder")
for i, l in enumerate(m.prediction_heads): hook(l, "fwd.heads")
hook(m.proposal_head, "fwd.heads"); hook(m.points_obj_cls, "fwd.objcls")
def step():
    with T("fwd.total"):
        ep = model(inputs)
    with T("loss"):
        loss = surrogate_loss(ep, targets)
    with T("bwd"):
        opt.zero_grad(set_to_none=True); loss.backward()
    with T("clip+adamw"):
        torch.nn.utils.clip_grad_norm_([p for g in opt.param_groups for p in g["params"]], 0.1); opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); regions.clear()
t0 = time.perf_counter(); step(); torch.cuda.synchronize(); wall = time.perf_counter() - t0
import collections
agg = collections.OrderedDict()
for r in regions:
    a = agg.setdefault(r.name, [0.0, 0.0]); a[0] += r.e0.elapsed_time(r.e1); a[1] += (r.t1 - r.t0) * 1e3
print(f"backend={backend} step wall {wall*1e3:.1f} ms")
for k, (g, h) in agg.items(): print(f"{k:14s} gpu-span {g:8.2f} ms   host {h:8.2f} ms")
