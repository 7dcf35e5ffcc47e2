"""attention core at 1024 x 1024 for several batch sizes: time per workgroup vs the number of workgroups
(768 = one full round of 3 resident workgroups per CU; 1024 = the bench's grid)"""
import sys; sys.path.insert(0, '.')
import torch
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
for B in (2, 4, 6, 8, 10, 12, 16, 24):
    q = torch.randn(B, Lq, E, device='cuda'); k = torch.randn(B, Lk, E, device='cuda'); v = torch.randn(B, Lk, E, device='cuda')
    out = torch.empty_like(q); lse = torch.empty(B, H, Lq, device='cuda'); do = torch.randn_like(q)
    dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v); delta = torch.empty(B, H, Lq, device='cuda')
    ctr = fa.rng_counter(q.device).data_ptr(); st = lambda: torch.cuda.current_stream().cuda_stream
    p = 0.1
    f = lambda: lib.butd_attention_fwd(B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), None, out.data_ptr(), lse.data_ptr(), p, 7, ctr, st())
    b = lambda: lib.butd_attention_bwd(B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), None, out.data_ptr(), do.data_ptr(), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), 0, 0, 1.0, p, 7, ctr, st())
    tf, tb = tg(f), tg(b)
    wgs = 16 * H * B
    fl = 4.0 * Lq * Lk * D * H * B
    print(f"B={B:2d} workgroups {wgs:5d}: fwd {tf:7.1f} us ({fl / tf / 1e6:5.1f} TF)  bwd {tb:7.1f} us ({2.5 * fl / tb / 1e6:5.1f} TF)  fwd us per 256 wgs {tf / wgs * 256:.1f}")
