"""host-side time of the phases of GraphedTrainStep.__call__ per step: is anything blocking on the GPU?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, synthetic_batch
args = bench.parse()
dev = torch.device("cuda", 0)
model, _ = bench.build_model(args, dev)
crit = bench.make_criterion(args)
step = GraphedTrainStep(model, FlatAdamW(model), criterion=crit)
batches = [synthetic_batch(args.batch, dev, seed=1184 + 37 * i, n_points=args.points, tokens=args.tokens) for i in range(4)]
for k in range(5):
    step(*batches[k % 4], next_inputs=batches[(k + 1) % 4][0])
torch.cuda.synchronize()
# wrap phases
import types
T = {}
def timed(obj, name, label):
    f = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[label] = T.get(label, 0.0) + time.perf_counter() - t0; return r
    setattr(obj, name, w)
timed(step, "_tokenize", "tokenize(next)")
timed(step, "_copy_in", "copy_in")
timed(crit, "prepare", "criterion.prepare")
timed(step.optimizer, "sync_hyper", "sync_hyper")
for g in ("g_fwd_bwd", "g_update", "g_stage1", "g_stage2"):
    if hasattr(step._slot, g):
        timed(getattr(step._slot, g), "replay", g + ".replay")
N = 20
t0 = time.perf_counter()
for k in range(5, 5 + N):
    step(*batches[k % 4], next_inputs=batches[(k + 1) % 4][0])
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host loop {t_host / N * 1e3:.3f} ms/step, with final sync {t_all / N * 1e3:.3f} ms/step")
for k, v in sorted(T.items(), key=lambda kv: -kv[1]): print(f"  {k:22s} {v / N * 1e3:8.3f} ms/step")
