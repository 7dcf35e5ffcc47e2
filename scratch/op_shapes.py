"""stock aten elementwise ops of one forward+loss+backward, by op / phase / shape / python site (fresh .grad: as in the graph step)"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.utils._python_dispatch import TorchDispatchMode
from butd_detr_amd.train_step import synthetic_batch
from butd_detr_amd.fused_attention import ZeroArena
args = bench.parse()
device = torch.device("cuda", 0)
model, backend = bench.build_model(args, device)
inputs, targets = synthetic_batch(args.batch, device, n_points=args.points, tokens=args.tokens, rank=0)
inputs["text_encoder_output"] = model.encode_text(model.tokenize(inputs))     # as prefetched in the graph step
crit = bench.make_criterion(args)
targets = crit.prepare(targets)
arena = ZeroArena(device)
with arena:
    loss = crit(model(inputs), targets); loss.backward()     # warm
for p in model.parameters(): p.grad = None
KEEP = ("add", "copy_", "clone", "_to_copy", "zeros", "fill_", "zero_", "mul", "div", "cat", "stack", "sum", "contiguous", "where", "index")
ALL = os.environ.get("ALL_OPS") == "1"
SKIP = ("view", "_unsafe_view", "t.", "transpose", "detach", "alias", "expand", "slice", "select", "unsqueeze", "squeeze",
        "permute", "as_strided", "empty", "reshape", "unbind", "split", "_local_scalar", "is_", "size", "stride", "new_empty",
        "lift", "sym_", "record_stream", "_reshape_alias")
class Log(TorchDispatchMode):
    def __init__(self): super().__init__(); self.rows = collections.Counter(); self.phase = "fwd"
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        if ALL and not name.startswith(SKIP) or name.split(".")[0] in KEEP:
            shp = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), ())
            site = "autograd"
            try:
                node = torch._C._current_autograd_node()
                if node is not None:
                    site = "autograd:" + node.name()
            except Exception:
                pass
            for fr in reversed(traceback.extract_stack(limit=40)):
                if "/butd_detr_amd/" in fr.filename and "op_shapes" not in fr.filename:
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno}"; break
            self.rows[(self.phase, name, shp, site)] += 1
        return func(*args, **(kwargs or {}))
log = Log()
arena.reset()
with arena, log:
    ep = model(inputs)
    loss = crit(ep, targets)
    log.phase = "bwd"
    loss.backward()
torch.cuda.synchronize()
tot = collections.Counter()
for (ph, name, shp, site), n in log.rows.items(): tot[(ph, name.split(".")[0])] += n
print("totals:", dict(tot))
bysite = collections.Counter()
for (ph, name, shp, site), n in log.rows.items(): bysite[(ph, site)] += n
if ALL:
    print("by site:")
    for (ph, site), n in sorted(bysite.items(), key=lambda kv: -kv[1])[:100]: print(f"{n:4d} {ph} {site}")
for (ph, name, shp, site), n in sorted(log.rows.items(), key=lambda kv: -kv[1])[:400]:
    print(f"{n:4d} {ph} {name:22s} {str(shp):24s} {site}")
