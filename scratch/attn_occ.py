"""forward attention time vs batch (blocks per CU = B/2 at 1024 queries, 8 heads): do co-resident workgroups overlap?"""
import sys; sys.path.insert(0, '.')
import os, torch
from butd_detr_amd import fused_attention as fa, _hiplib
lib = _hiplib.load()
def tg(fn, reps=10):
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
H, D = 8, 36; E = H * D
Lq = Lk = 1024
for B in (1, 2, 4, 6, 8, 12, 16):
    q = torch.randn(B, Lq, E, device='cuda'); k = torch.randn(B, Lk, E, device='cuda'); v = torch.randn(B, Lk, E, device='cuda')
    out = torch.empty_like(q); lse = torch.empty(B, H, Lq, device='cuda')
    ctr = fa.rng_counter(q.device).data_ptr(); st = lambda: torch.cuda.current_stream().cuda_stream
    f = lambda: lib.butd_attention_fwd(B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), None, out.data_ptr(), lse.data_ptr(), 0.1, 7, ctr, st())
    print(f"B={B:2d} blocks={B*H*16:5d} ({B*H*16/256:.1f}/CU): fwd {tg(f):6.1f} us")
