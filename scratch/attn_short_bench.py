"""round 4: one-kernel backward for short key sets vs the two-kernel backward (graph-replay timing, 8 x 8 heads x 36)."""
import sys; sys.path.insert(0, '.')
import torch
from butd_detr_amd import fused_attention as fa, _hiplib
lib = _hiplib.load()


def tg(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g):
            for _ in range(reps): fn()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


B, H, D = 8, 8, 36; E = H * D
for Lq, Lk in ((256, 80), (256, 132), (1024, 80), (1024, 132), (80, 80)):
    q = torch.randn(B, Lq, E, device='cuda'); k = torch.randn(B, Lk, E, device='cuda'); v = torch.randn(B, Lk, E, device='cuda')
    out = torch.empty_like(q); lse = torch.empty(B, H, Lq, device='cuda'); do = torch.randn_like(q)
    dq = torch.empty_like(q); dk = torch.zeros_like(k); dv = torch.zeros_like(v); delta = torch.empty(B, H, Lq, device='cuda')
    ctr = fa.rng_counter(q.device).data_ptr(); st = lambda: torch.cuda.current_stream().cuda_stream
    p = 0.1
    lib.butd_attention_fwd(B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), None, out.data_ptr(), lse.data_ptr(), p, 7, ctr, st())
    args = (B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), None, out.data_ptr(), do.data_ptr(), lse.data_ptr(),
            delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), 0, 0, 1.0, p, 7, ctr)
    two = lambda: lib.butd_attention_bwd(*args, st())
    one = lambda: lib.butd_attention_bwd_short_keys(*args, st())
    print(f"Lq={Lq} Lk={Lk}: two kernels {tg(two):6.1f} us   one kernel {tg(one):6.1f} us", flush=True)
