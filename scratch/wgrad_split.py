import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from butd_detr_amd import _hiplib, fused_attention as fa
lib = _hiplib.load(); dev = torch.device("cuda", 0)
def tg(fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g):
            for _ in range(reps): fn()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
E = 288
for M in (8192, 2048):
    dy = [torch.randn(M, E, device=dev) for _ in range(3)]; x = [torch.randn(M, E, device=dev) for _ in range(3)]
    dw = [torch.zeros(E, E, device=dev) for _ in range(3)]; db = [torch.zeros(E, device=dev) for _ in range(3)]
    w = torch.randn(E, E, device=dev); dx = torch.empty(M, E, device=dev)
    for split in (2, 4, 8, 16, 32):
        if split > M // 64: continue
        def wg(i):
            return fa._problem(dy[i], x[i], dw[i], E, E, M, (1, E), (1, E), E, bias_grad=db[i], ones_col=True, accumulate=True, split_k=split)
        row = []
        for tile in ((0, 0), (96, -32), (32, -32), (64, -64), (96, -96) if False else (128, -64)):
            lib.butd_gemm_set_tile(*tile)
            p3 = [wg(i) for i in range(3)]
            t3 = tg(lambda: fa._gemm(p3, dy[0]))
            p2 = [fa._dgrad(dy[0], w, dx, M, E, E), wg(0)]
            t2 = tg(lambda: fa._gemm(p2, dy[0]))
            row.append("%s: w3 %.1f dw %.1f" % ("auto" if tile == (0, 0) else "%dx%d" % (tile[0], -tile[1]), t3, t2))
        lib.butd_gemm_set_tile(0, 0)
        print("M=%d split=%2d  " % (M, split) + " | ".join(row), flush=True)
