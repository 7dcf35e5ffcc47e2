"""forward attention kernel time per ablation build (scratch/build_abl.sh attention_ops ...): BUTD_HIP_LIB picks the lib"""
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
B, H, D = 8, 8, 36; E = H * D
row = [os.path.basename(os.environ.get("BUTD_HIP_LIB", "product"))]
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
for Lq, Lk in ((1024, 1024), (256, 1024)):
    q = torch.randn(B, Lq, E, device='cuda'); k = torch.randn(B, Lk, E, device='cuda'); v = torch.randn(B, Lk, E, device='cuda')
    out = torch.empty_like(q); lse = torch.empty(B, H, Lq, device='cuda'); do = torch.randn_like(q)
    dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v); delta = torch.empty(B, H, Lq, device='cuda')
    ctr = fa.rng_counter(q.device).data_ptr(); st = lambda: torch.cuda.current_stream().cuda_stream
    p = 0.1
    f = lambda: lib.butd_attention_fwd(B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), None, out.data_ptr(), lse.data_ptr(), p, 7, ctr, st())
    bw = lambda: lib.butd_attention_bwd(B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), None, out.data_ptr(), do.data_ptr(), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), 0, 0, 1.0, p, 7, ctr, st())
    row.append(f"{Lq}x{Lk}: fwd {tg(f):6.1f} us" + (f" bwd {tg(bw):6.1f} us" if which == "all" else ""))
print("  ".join(row))
