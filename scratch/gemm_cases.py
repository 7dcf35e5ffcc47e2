"""Graph-replay timing of the step's representative butd_gemm_grouped launches under every tile of the
kernel's menu (butd_gemm_set_tile) and under the built-in choice.  One table row per (case, tile)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from butd_detr_amd import _hiplib, fused_attention as fa
lib = _hiplib.load()
dev = torch.device("cuda", 0)
TILES = [(0, 0), (32, 32), (32, -32), (64, 64), (64, -64), (32, 96), (32, -96), (64, 96), (64, -96), (96, 32), (96, -32), (128, 64), (128, -64), (128, 96), (128, -96)]
if len(sys.argv) > 1 and sys.argv[1] == "auto":
    TILES = [(0, 0)]
if os.environ.get("TILES"):
    TILES = [tuple(int(v) for v in t.split("x")) for t in os.environ["TILES"].split(",")]

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

keep = []
def R(*s):
    t = torch.randn(*s, device=dev); keep.append(t); return t
def Z(*s):
    t = torch.zeros(*s, device=dev); keep.append(t); return t
def fwd(M, N, K, aff=False, stats=False):
    kw = {}
    if aff: kw["a_affine"] = (R(K).abs() + 0.5, R(K))
    return fa._fwd(R(M, K), R(N, K), Z(M, N), M, N, K, bias=R(N), **kw)
def dgrad(M, N, K):
    return fa._dgrad(R(M, N), R(N, K), Z(M, K), M, N, K)
def wgrad(M, N, K, bias=True, baff=False):
    kw = {}
    if baff: kw["b_affine"] = (R(K).abs() + 0.5, R(K))
    return fa._wgrad(R(M, N), R(M, K), Z(N, K), Z(N) if bias else None, M, N, K, **kw)

E = 288
CASES = [
    ("fwd 3x(8192,288,288)", lambda: [fwd(8192, E, E) for _ in range(3)]),
    ("fwd 1x(8192,288,288)", lambda: [fwd(8192, E, E)]),
    ("fwd 3x(2048,288,288)", lambda: [fwd(2048, E, E) for _ in range(3)]),
    ("fwd 1x(2048,288,288)", lambda: [fwd(2048, E, E)]),
    ("fwd (2048)+2x(8192)", lambda: [fwd(2048, E, E), fwd(8192, E, E), fwd(8192, E, E)]),
    ("fwd (2048)+2x(640)", lambda: [fwd(2048, E, E), fwd(640, E, E), fwd(640, E, E)]),
    ("fwd (8192,256,288)", lambda: [fwd(8192, 256, E)]),
    ("fwd (8192,288,256)", lambda: [fwd(8192, E, 256)]),
    ("fwd (2048,256,288)", lambda: [fwd(2048, 256, E)]),
    ("dgrad+wgrad 8192", lambda: [dgrad(8192, E, E), wgrad(8192, E, E)]),
    ("dgrad+wgrad 2048", lambda: [dgrad(2048, E, E), wgrad(2048, E, E)]),
    ("dgrad 3x8192", lambda: [dgrad(8192, E, E) for _ in range(3)]),
    ("dgrad 3x2048", lambda: [dgrad(2048, E, E) for _ in range(3)]),
    ("wgrad 3x8192", lambda: [wgrad(8192, E, E) for _ in range(3)]),
    ("wgrad 3x2048", lambda: [wgrad(2048, E, E) for _ in range(3)]),
    ("wgrad 2048+2x8192", lambda: [wgrad(2048, E, E), wgrad(8192, E, E), wgrad(8192, E, E)]),
    ("dgrad(8192,576->288)+wgrad(576x288)", lambda: [dgrad(8192, 2 * E, E), wgrad(8192, 2 * E, E)]),
    ("SA fwd (1M,128,64) aff", lambda: [fwd(1 << 20, 128, 64, aff=True)]),
    ("SA fwd (1M,64,64) aff", lambda: [fwd(1 << 20, 64, 64, aff=True)]),
    ("SA fwd (256k,256,128) aff", lambda: [fwd(1 << 18, 256, 128, aff=True)]),
    ("SA fwd (256k,128,128) aff", lambda: [fwd(1 << 18, 128, 128, aff=True)]),
    ("SA fwd (256k,128,132)", lambda: [fwd(1 << 18, 128, 132)]),
    ("SA fwd (64k,256,128) aff", lambda: [fwd(1 << 16, 256, 128, aff=True)]),
    ("SA fwd (64k,128,260)", lambda: [fwd(1 << 16, 128, 260)]),
    ("SA bwd w(128,64,1M)+d(1M,64<-128)", lambda: [wgrad(1 << 20, 128, 64, bias=False, baff=True), dgrad(1 << 20, 128, 64)]),
    ("SA bwd w(64,64,1M)+d(1M,64<-64)", lambda: [wgrad(1 << 20, 64, 64, bias=False, baff=True), dgrad(1 << 20, 64, 64)]),
    ("SA bwd w(256,128,256k)+d", lambda: [wgrad(1 << 18, 256, 128, bias=False, baff=True), dgrad(1 << 18, 256, 128)]),
    ("SA bwd w(128,128,256k)+d", lambda: [wgrad(1 << 18, 128, 128, bias=False, baff=True), dgrad(1 << 18, 128, 128)]),
    ("SA bwd w(256,128,64k)+d", lambda: [wgrad(1 << 16, 256, 128, bias=False, baff=True), dgrad(1 << 16, 256, 128)]),
]
only = os.environ.get("CASES")
print("%-40s %s" % ("case", " ".join("%9s" % ("auto" if t == (0, 0) else "%dx%d%s" % (t[0], abs(t[1]), "/1" if t[1] < 0 else "")) for t in TILES)))
for name, mk in CASES:
    if only and only not in name:
        continue
    keep.clear()
    probs = mk()
    flops = sum(2.0 * p.M * p.N * p.K for p in probs)
    ref = keep[0]
    row = []
    for t in TILES:
        assert lib.butd_gemm_set_tile(*t) == 0
        us = tg(lambda: fa._gemm(probs, ref))
        row.append("%5.1f/%3.0f" % (us, flops / us / 1e6))
    lib.butd_gemm_set_tile(0, 0)
    print("%-40s %s   (us/TF)" % (name, " ".join(row)), flush=True)
    torch.cuda.empty_cache()
