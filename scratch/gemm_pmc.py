import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from butd_detr_amd import _hiplib, fused_attention as fa
lib = _hiplib.load()
dev = torch.device("cuda", 0)
tm, tn = (int(v) for v in os.environ.get("TILE", "64x64").split("x"))
M = int(os.environ.get("ROWS", "8192")); E = 288
xs = [torch.randn(M, E, device=dev) for _ in range(3)]; ws = [torch.randn(E, E, device=dev) for _ in range(3)]
ys = [torch.empty(M, E, device=dev) for _ in range(3)]; bs = [torch.randn(E, device=dev) for _ in range(3)]
probs = [fa._fwd(x, w, y, M, E, E, bias=b) for x, w, y, b in zip(xs, ws, ys, bs)]
lib.butd_gemm_set_tile(tm, tn)
for _ in range(5):
    fa._gemm(probs, xs[0])
torch.cuda.synchronize()
