"""Where the cycles of a key tile of attn_fwd_kernel go (one wave: wave 0 of workgroup 0), from s_memtime stamps compiled in
with -DATTN_PROF (scratch/build_abl.sh attention_ops prof "-DATTN_PROF"; BUTD_HIP_LIB=scratch/exp/libabl_prof.so).
The unit is what s_memtime counts on this part while the wave runs (compare the columns with each other, not with wall time)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from butd_detr_amd import fused_attention as fa, _hiplib
lib = _hiplib.load()
lib.butd_attention_prof_read.restype = ctypes.c_int
lib.butd_attention_prof_read.argtypes = [ctypes.c_void_p]
H, D = 8, 36
E = H * D
names = ["fetch issue", "S = K.Q^T (36 matrix)", "softmax + dropout", "P.V (48 matrix)", "commit + barrier"]
for (B, Lq, Lk) in ((2, 1024, 1024), (4, 1024, 1024), (6, 1024, 1024), (8, 1024, 1024), (8, 256, 1024)):
    q, k, v = (torch.randn(B, L, E, device="cuda") for L in (Lq, Lk, Lk))
    out, lse = torch.empty_like(q), torch.empty(B, H, Lq, device="cuda")
    ctr = fa.rng_counter(q.device).data_ptr()
    for p in (0.1, 0.0):
        st = torch.cuda.current_stream().cuda_stream
        call = lambda: lib.butd_attention_fwd(B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), None, out.data_ptr(),
                                              lse.data_ptr(), p, 7, ctr, st)
        call(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call()
        e1.record(); torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 8)()
        assert lib.butd_attention_prof_read(buf) == 0
        tiles = buf[5]
        per = [buf[i] / tiles for i in range(5)]
        print(f"B={B} {Lq}x{Lk} p={p}: {e0.elapsed_time(e1) / 10 * 1e3:7.1f} us/launch, workgroups {B * H * ((Lq + 63) // 64)}; "
              f"per key tile of one wave (s_memtime ticks): "
              + ", ".join(f"{n} {t:.1f}" for n, t in zip(names, per)) + f"; total {sum(per):.1f} ticks")
