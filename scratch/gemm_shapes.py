"""Per-launch gemm_kernel table of one eager training step: shapes, flags, time, TF."""
import os, sys, json, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from butd_detr_amd.train_step import make_optimizer, synthetic_batch, train_step as eager_step
from butd_detr_amd import fused_attention as fa
import butd_detr_amd.fused_sa as fsa

class A: pass
args = bench.parse()
device = torch.device("cuda", 0)
model, backend = bench.build_model(args, device)
inputs, targets = synthetic_batch(args.batch, device, n_points=args.points, tokens=args.tokens, rank=0)
opt = make_optimizer(model)
for _ in range(2):
    eager_step(model, opt, inputs, targets)
torch.cuda.synchronize()
stream = torch.cuda.current_stream()
records = []
orig = fa._gemm
def desc(p):
    return (p.M, p.N, p.K, int(p.lda_k == 1), int(p.ldb_k == 1), int(bool(p.a2)), int(p.ones_col), int(p.split_k),
            int(p.accumulate), int(bool(p.a_chan_scale)), int(bool(p.b_chan_scale)))
def timed(problems, ref):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream); orig(problems, ref); e1.record(stream)
    records.append((tuple(desc(p) for p in problems), sum(2.0 * p.M * p.N * p.K for p in problems), e0, e1))
fa._gemm = timed; fsa._gemm = timed
eager_step(model, opt, inputs, targets)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for d, fl, e0, e1 in records:
    a = agg.setdefault(d, [0, 0.0, 0.0]); a[0] += 1; a[1] += fl; a[2] += e0.elapsed_time(e1)
rows = sorted(agg.items(), key=lambda kv: -kv[1][2])
tot = sum(v[2] for v in agg.values())
print("total gemm ms %.3f launches %d" % (tot, len(records)))
print("cols: (M,N,K,a_kc,b_kc,a2,ones,splitk,acc,a_aff,b_aff)")
for d, (n, fl, ms) in rows:
    print("%3d x  %8.1f us/launch  %6.1f TF  tot %6.3f ms  %s" % (n, ms / n * 1e3, fl / (ms * 1e-3) / 1e12, ms, d))
