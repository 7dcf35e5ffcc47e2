"""The one-pass attention backward for long key sets (butd_attention_bwd_long_keys) vs the two-kernel walk:
results side by side (packed gradient ld = 3E, dropout 0.1, dq_scale 0.5) and graph-replayed times."""
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

B, H, D = 8, 8, 36; E = H * D
dev = torch.device("cuda", 0)
ctr = fa.rng_counter(dev).data_ptr()
st = lambda: torch.cuda.current_stream().cuda_stream
import os
SHAPES = ((1024, 1024, False), (256, 1024, False), (80, 1024, False), (1024, 512, False), (1000, 1000, True), (2048, 2048, False))
if os.environ.get("SHAPES") == "short":
    SHAPES = ((256, 256, False), (256, 80, True), (256, 132, True), (1024, 132, True), (1024, 80, True), (80, 80, True))
# PLANS: "keys per workgroup:query splits" (0 = the library's rule), e.g. PLANS=0:0,64:1,64:4,64:8
CHUNKS = [tuple(int(x) for x in c.split(":")) for c in os.environ.get("PLANS", "0:0").split(",")]
for Lq, Lk, masked in SHAPES:
    torch.manual_seed(Lq + Lk)
    q, do = torch.randn(B, Lq, E, device=dev), torch.randn(B, Lq, E, device=dev)
    k, v = torch.randn(B, Lk, E, device=dev), torch.randn(B, Lk, E, device=dev)
    mask = None
    if masked:
        mask = torch.zeros(B, Lk, dtype=torch.uint8, device=dev)
        mask[:, Lk - min(77, Lk // 4):] = 1
        if Lk >= 512:
            mask[1, 100:400] = 1
    mp = mask.data_ptr() if mask is not None else None
    o, lse, delta = torch.empty_like(q), torch.empty(B, H, Lq, device=dev), torch.empty(B, H, Lq, device=dev)
    for p in ((0.1,) if os.environ.get("SHAPES") == "short" else (0.0, 0.1)):
        lib.butd_attention_fwd(B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), mp, o.data_ptr(), lse.data_ptr(), p, 5, ctr, st())
        G, Gq = torch.empty(B, Lk, 3 * E, device=dev), torch.empty(B, Lq, 3 * E, device=dev)
        args = lambda: (B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), mp, o.data_ptr(), do.data_ptr(), lse.data_ptr())
        outs = lambda: (Gq.data_ptr(), G.data_ptr() + 4 * E, G.data_ptr() + 8 * E, 3 * E, 3 * E, 0.5, p, 5, ctr)
        two = lambda: lib.butd_attention_bwd(*args(), delta.data_ptr(), *outs(), st())
        assert two() == 0
        torch.cuda.synchronize(); ref, refq = G.clone(), Gq.clone()
        line = f"  {Lq:5d} x {Lk:5d} p={p} mask={int(masked)}: two kernels {tg(two):7.1f} us |"
        if Lk <= 144:
            G.zero_()
            short = lambda: lib.butd_attention_bwd_short_keys(*args(), delta.data_ptr(), *outs(), st())
            line += f" short-key kernel {tg(short):7.1f} us |"
        for chunk in CHUNKS:
            lib.butd_attention_bwd_long_keys_set_chunk(*chunk)
            need = int(lib.butd_attention_bwd_long_keys_scratch(B, H, Lq, Lk, D, 3 * E))
            if need < 0:
                line += f" {chunk[0]}:{chunk[1]} not served |"
                continue
            ws = torch.empty(max(need, 1), device=dev)
            one = lambda: lib.butd_attention_bwd_long_keys(*args(), *outs(), ws.data_ptr(), need, st())
            G.fill_(float("nan")); Gq.fill_(float("nan"))
            assert one() == 0
            torch.cuda.synchronize()
            worst = 0.0
            for name, t, rt, lo in (("dq", Gq, refq, 0), ("dk", G, ref, E), ("dv", G, ref, 2 * E)):
                a, r = t[:, :, lo:lo + E].double(), rt[:, :, lo:lo + E].double()
                worst = max(worst, float((a - r).abs().max() / r.abs().max()))
            line += f" {chunk[0]}:{chunk[1]} {tg(one):6.1f} ({worst:.0e}) |"
        lib.butd_attention_bwd_long_keys_set_chunk(0, 0)
        nb = int(lib.butd_attention_bwd_long_keys_bf16_scratch(B, H, Lq, Lk, D, 3 * E))
        if nb >= 0 and os.environ.get("BF16") == "1":       # the bf16 entry points: two kernels | one pass
            G.zero_(); Gq.zero_()
            wsb = torch.empty(max(nb, 1), device=dev)
            two_h = lambda: lib.butd_attention_bwd_bf16(*args(), delta.data_ptr(), *outs(), st())
            one_h = lambda: lib.butd_attention_bwd_long_keys_bf16(*args(), *outs(), wsb.data_ptr(), nb, st())
            assert two_h() == 0 and one_h() == 0
            line += f" bf16: two kernels {tg(two_h):6.1f} | one pass {tg(one_h):6.1f} |"
        print(line, flush=True)
