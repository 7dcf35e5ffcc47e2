"""round 4: the row-panel chain kernel alone (graph-replay timing) against the launches it replaces."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from butd_detr_amd import fused_attention as fa, _hiplib
lib = _hiplib.load()
dev = torch.device("cuda", 0)
E, Fh = 288, 256


def timed(fn, iters=50):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


for rows in (2048, 8192, 640):
    torch.manual_seed(0)
    att, x, pos = (torch.randn(rows, E, device=dev) for _ in range(3))
    w = [torch.randn(E, E, device=dev) / 17 for _ in range(4)]
    b = [torch.randn(E, device=dev) * 0.1 for _ in range(4)]
    w1, b1 = torch.randn(Fh, E, device=dev) / 17, torch.randn(Fh, device=dev) * 0.1
    w2, b2 = torch.randn(E, Fh, device=dev) / 16, torch.randn(E, device=dev) * 0.1
    g1, be1, g2, be2 = (torch.ones(E, device=dev) for _ in range(4))
    outs = [torch.empty(rows, E, device=dev) for _ in range(8)]
    h = torch.empty(rows, Fh, device=dev)
    st = [torch.empty(rows, device=dev) for _ in range(4)]
    p = 0.1
    ln1 = (g1, be1, 1e-5, st[0], st[1]); ln2 = (g2, be2, 1e-5, st[2], st[3])

    def tail1():      # out-proj + LN only
        fa._panel(rows, att, E, [fa._stage(w[0], b[0], E, E, 0, 1, pre=outs[0], drop=(p, 1), ln=ln1, res=x, out=outs[1])], 2, att)

    def tail2():      # + pos + next q
        fa._panel(rows, att, E, [fa._stage(w[0], b[0], E, E, 0, 1, pre=outs[0], drop=(p, 1), ln=ln1, res=x, out=outs[1], pos=pos, out_pos=outs[2], pos_buf=2),
                                 fa._stage(w[1], b[1], E, E, 2, -1, scale=1 / 6, out=outs[3])], 3, att)

    def tail4():      # + q, k, v
        fa._panel(rows, att, E, [fa._stage(w[0], b[0], E, E, 0, 1, pre=outs[0], drop=(p, 1), ln=ln1, res=x, out=outs[1], pos=pos, out_pos=outs[2], pos_buf=2),
                                 fa._stage(w[1], b[1], E, E, 2, -1, scale=1 / 6, out=outs[3]),
                                 fa._stage(w[2], b[2], E, E, 1, -1, out=outs[4]),
                                 fa._stage(w[3], b[3], E, E, 1, -1, out=outs[5])], 3, att)

    def tailffn():
        fa._panel(rows, att, E, [fa._stage(w[0], b[0], E, E, 0, 1, pre=outs[0], drop=(p, 1), ln=ln1, res=x, out=outs[1]),
                                 fa._stage(w1, b1, Fh, E, 1, 2, relu=True, drop=(p, 2), out=h),
                                 fa._stage(w2, b2, E, Fh, 2, 0, pre=outs[6], drop=(p, 3), ln=ln2, res_buf=1, out=outs[7])], 3, att)

    def qkv():
        fa._panel(rows, x, E, [fa._stage(w[1], b[1], E, E, 1, -1, scale=1 / 6, out=outs[3]),
                               fa._stage(w[2], b[2], E, E, 1, -1, out=outs[4]),
                               fa._stage(w[3], b[3], E, E, 0, -1, out=outs[5])], 2, att, in_pos=pos, in_sum=outs[2])

    def old_tail():   # out-proj GEMM + LN kernel
        fa._gemm([fa._fwd(att, w[0], outs[0], rows, E, E, bias=b[0])], att)
        lib.butd_add_dropout_layernorm_fwd_pos(rows, E, outs[0].data_ptr(), x.data_ptr(), g1.data_ptr(), be1.data_ptr(), 1e-5,
                                               outs[1].data_ptr(), st[0].data_ptr(), st[1].data_ptr(), p, 1,
                                               fa.rng_counter(dev).data_ptr(), pos.data_ptr(), outs[2].data_ptr(), fa._stream(att))

    def old_q():
        fa._gemm([fa._fwd(outs[2], w[1], outs[3], rows, E, E, bias=b[1], scale=1 / 6)], att)

    def old_qkv():
        fa._gemm([fa._fwd(outs[2], w[1], outs[3], rows, E, E, bias=b[1], scale=1 / 6),
                  fa._fwd(outs[2], w[2], outs[4], rows, E, E, bias=b[2]),
                  fa._fwd(x, w[3], outs[5], rows, E, E, bias=b[3])], att)

    def old_ffn():
        fa._gemm([fa._fwd(outs[1], w1, h, rows, Fh, E, bias=b1, relu=True, dropout_p=p, site=2)], att)
        fa._gemm([fa._fwd(h, w2, outs[6], rows, E, Fh, bias=b2)], att)
        lib.butd_add_dropout_layernorm_fwd(rows, E, outs[6].data_ptr(), outs[1].data_ptr(), g2.data_ptr(), be2.data_ptr(), 1e-5,
                                           outs[7].data_ptr(), st[2].data_ptr(), st[3].data_ptr(), p, 3,
                                           fa.rng_counter(dev).data_ptr(), fa._stream(att))

    for forced in (16, 32):
        lib.butd_panel_set_rows(forced)
        print(f"rows {rows} R={forced}: " + "  ".join(f"{n} {timed(f):6.1f}us" for n, f in
              (("tail1", tail1), ("tail2", tail2), ("tail4", tail4), ("tail+ffn", tailffn), ("qkv", qkv))), flush=True)
    lib.butd_panel_set_rows(0)
    print(f"rows {rows} per-op launches: " + "  ".join(f"{n} {timed(f):6.1f}us" for n, f in
          (("outproj+ln", old_tail), ("q", old_q), ("qkv", old_qkv), ("ffn(3)", old_ffn))), flush=True)
