"""which python lines issue the stock aten ops of one forward+loss (+ backward's python-visible part)"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.utils._python_dispatch import TorchDispatchMode
from butd_detr_amd.train_step import synthetic_batch
args = bench.parse()
device = torch.device("cuda", 0)
model, backend = bench.build_model(args, device)
inputs, targets = synthetic_batch(args.batch, device, n_points=args.points, tokens=args.tokens, rank=0)
crit = bench.make_criterion(args)
targets = crit.prepare(targets)
loss = crit(model(inputs), targets); loss.backward()     # warm
SKIP = ("aten.view", "aten._unsafe_view", "aten.t.", "aten.transpose", "aten.detach", "aten.alias", "aten.expand", "aten.slice",
        "aten.select", "aten.unsqueeze", "aten.squeeze", "aten.permute", "aten.as_strided", "aten.empty", "aten.reshape", "aten.unbind",
        "aten.split", "aten._local_scalar", "aten.is_", "aten.size", "aten.stride", "aten.new_empty", "aten.lift", "aten.sym_")
class Log(TorchDispatchMode):
    def __init__(self): super().__init__(); self.rows = collections.Counter(); self.phase = "fwd"
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            shp = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), ())
            numel = 1
            for s in shp: numel *= s
            site = "?"
            for fr in reversed(traceback.extract_stack(limit=40)):
                if "/butd_detr_amd/" in fr.filename and "op_sites" not in fr.filename:
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno}"; break
            tag = shp if any(k in name for k in ("copy", "clone", "contiguous", "fill", "zeros", "cat", "stack")) and numel >= 100000 else (numel >= 100000)
            self.rows[(self.phase, name.replace("aten.", ""), site, tag)] += 1
        return func(*args, **(kwargs or {}))
log = Log()
with log:
    ep = model(inputs)
    loss = crit(ep, targets)
    log.phase = "bwd"
    loss.backward()
torch.cuda.synchronize()
tot = collections.Counter()
for (ph, name, site, big), n in log.rows.items(): tot[ph] += n
print("ops:", dict(tot))
for (ph, name, site, big), n in sorted(log.rows.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d} {ph} {str(big) if not isinstance(big, bool) else ('BIG' if big else '   '):22s} {name:30s} {site}")
