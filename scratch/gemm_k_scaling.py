"""How much of a small grouped-GEMM launch is its K loop?  2048 x 288 x K (and 8192 rows) for K = 32 .. 1152, graph-replay timing:
the intercept is launch + prologue + epilogue, the slope the cost of a 32-deep slab iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from butd_detr_amd import fused_attention as fa
dev = torch.device("cuda", 0)
def tg(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g):
            for _ in range(reps): fn()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
for M in (2048, 8192, 640):
    row = []
    for K in (32, 96, 160, 288, 576, 1152):
        x, w, y = torch.randn(M, K, device=dev), torch.randn(288, K, device=dev), torch.empty(M, 288, device=dev)
        row.append((K, tg(lambda: fa._gemm([fa._fwd(x, w, y, M, 288, K)], x))))
    print(f"M={M} N=288: " + "  ".join(f"K={k}: {t:.2f} us" for k, t in row) + f"   slope {(row[-1][1] - row[0][1]) / ((1152 - 32) / 32):.3f} us per slab")
