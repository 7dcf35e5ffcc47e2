"""round 6: which Python call sites issue the stock-torch element-wise launches of one eager training step (bench
configuration): counts of add / cat / contiguous-copy / clone / zeros / fill per butd_detr_amd file:line, forward only
(autograd's own gradient accumulations have no Python frame: they are the rest of the trace's `add` launches)."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, torch
import bench
from butd_detr_amd.train_step import FlatAdamW, HungarianCriterion, synthetic_batch, train_step, make_optimizer
dev = torch.device("cuda", 0)
args = argparse.Namespace(backend="auto", queries=256, points=50000, tokens=80, encoder_layers=3)
model, _ = bench.build_model(args, dev)
crit = HungarianCriterion()
inputs, targets = synthetic_batch(8, dev, seed=1184, n_points=50000, tokens=80)
targets = crit.prepare(targets)
opt = make_optimizer(model)
train_step(model, opt, inputs, targets, criterion=crit)
torch.cuda.synchronize()
sites = collections.Counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "butd_detr_amd" in fr.filename and "r6_stock_sites" not in fr.filename:
            return f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno}"
    return "?"


def wrap(owner, name, tag, pred=lambda *a, **k: True):
    orig = getattr(owner, name)

    def f(*a, **k):
        t = next((x for x in a if torch.is_tensor(x)), None)
        if (t is None or t.is_cuda) and pred(*a, **k):
            sites[(tag, site())] += 1
        return orig(*a, **k)
    setattr(owner, name, f)


wrap(torch.Tensor, "__add__", "add"); wrap(torch.Tensor, "__radd__", "add"); wrap(torch, "add", "add")
wrap(torch.Tensor, "__sub__", "sub"); wrap(torch.Tensor, "__mul__", "mul"); wrap(torch.Tensor, "__truediv__", "div")
wrap(torch, "cat", "cat"); wrap(torch, "stack", "stack")
wrap(torch.Tensor, "contiguous", "contiguous(copy)", lambda t, *a, **k: not t.is_contiguous())
wrap(torch.Tensor, "clone", "clone"); wrap(torch.Tensor, "copy_", "copy_")
wrap(torch, "zeros", "zeros"); wrap(torch, "zeros_like", "zeros"); wrap(torch, "full", "full"); wrap(torch, "ones", "ones")
wrap(torch.Tensor, "fill_", "fill_"); wrap(torch.Tensor, "zero_", "zero_")
wrap(torch.Tensor, "float", "float()", lambda t, *a, **k: t.dtype != torch.float32)
wrap(torch.Tensor, "to", "to()")
train_step(model, opt, inputs, targets, criterion=crit)
torch.cuda.synchronize()
for (tag, where), n in sorted(sites.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{n:4d}  {tag:18s} {where}")
