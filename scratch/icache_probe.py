"""Do small launches pay for a cold instruction cache?  The same 2048 x 288 x 288 product (a) 20 times in a row, (b) alternating
with two OTHER kernels (an attention forward and a LayerNorm forward) -- time of the product = (b) - the others alone.
In the step every launch is a different kernel than the one before it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from butd_detr_amd import _hiplib, fused_attention as fa
lib = _hiplib.load()
dev = torch.device("cuda", 0)
def tg(fn, reps=1):
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
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best
M, E = 2048, 288
x, w, y = torch.randn(M, E, device=dev), torch.randn(E, E, device=dev), torch.empty(M, E, device=dev)
B, H, D, Lq, Lk = 8, 8, 36, 256, 80
q, k, v = (torch.randn(B, L, E, device=dev) for L in (Lq, Lk, Lk))
out, lse = torch.empty_like(q), torch.empty(B, H, Lq, device=dev)
ctr = fa.rng_counter(dev).data_ptr()
st = lambda: torch.cuda.current_stream().cuda_stream
gamma, beta = torch.ones(E, device=dev), torch.zeros(E, device=dev)
yl, mean, rstd = torch.empty(M, E, device=dev), torch.empty(M, device=dev), torch.empty(M, device=dev)
gemm = lambda: fa._gemm([fa._fwd(x, w, y, M, E, E)], x)
attn = lambda: lib.butd_attention_fwd(B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), None, out.data_ptr(), lse.data_ptr(), 0.1, 7, ctr, st())
ln = lambda: lib.butd_add_dropout_layernorm_fwd(M, E, y.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, yl.data_ptr(), mean.data_ptr(), rstd.data_ptr(), 0.1, 9, ctr, st())
N = 20
t_g = tg(lambda: [gemm() for _ in range(N)]) / N
t_a = tg(lambda: [attn() for _ in range(N)]) / N
t_l = tg(lambda: [ln() for _ in range(N)]) / N
t_mix = tg(lambda: [(gemm(), attn(), ln()) for _ in range(N)]) / N
t_al = tg(lambda: [(attn(), ln()) for _ in range(N)]) / N
print(f"same kernel back to back: gemm 2048x288x288 {t_g:.2f} us, attention fwd 256x80 {t_a:.2f} us, layernorm fwd {t_l:.2f} us; sum {t_g + t_a + t_l:.2f}")
print(f"interleaved gemm, attention, layernorm: {t_mix:.2f} us per triple (+{t_mix - (t_g + t_a + t_l):.2f} us vs the sum); attention + layernorm alternating: {t_al:.2f} (+{t_al - t_a - t_l:.2f})")
