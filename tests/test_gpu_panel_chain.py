"""-m gpu: the row-panel chain kernels (include/butd_panel.h) -- out-projection + dropout + residual + LayerNorm
(+ pos) + projections of the result + the FFN in ONE launch -- against (i) the plain-torch maths
(models/encoder_decoder_layers.py:75-124, 340-406; tolerance 1e-3 of the scale, north_star) and (ii) the one-launch-per-
operator path of round 3 under the same dropout counter (same masks: equal to rounding)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from butd_detr_amd import attention_blocks as ab
    from butd_detr_amd import fused_attention as fa
    from butd_detr_amd.encoder_decoder_layers import MultiheadAttention, _ffn
    prev = fa.set_panel_chain(True)      # (off by default: measured slower in the step, profiles/r04_panel_chain.txt)
    yield ab, fa, MultiheadAttention, _ffn
    fa.set_panel_chain(prev)


def _close(a, b, tol=1e-3, name="", frac=0.0):
    a, b = a.detach().float().cpu().numpy(), b.detach().float().cpu().numpy()
    scale = max(np.abs(b).max(), 1e-6)
    if frac:        # a pre-activation within rounding of zero may take the other side of a ReLU gate
        bad = np.abs(a - b) / scale > tol
        assert bad.mean() <= frac and np.abs(a - b).max() / scale < 50 * tol, (name, bad.sum(), np.abs(a - b).max() / scale)
        return
    np.testing.assert_allclose(a / scale, b / scale, rtol=0, atol=tol, err_msg=name)


def _rand_ln(E):
    n = torch.nn.LayerNorm(E).cuda()
    with torch.no_grad():
        n.weight.uniform_(0.7, 1.3)
        n.bias.uniform_(-0.2, 0.2)
    return n


@pytest.mark.parametrize("rows,force", [(2048, 0), (8192, 0), (640, 0), (100, 16), (100, 32), (2048, 32)])
def test_raw_chain_three_stages_vs_torch(mods, rows, force):
    """tail + FFN as three stages, straight through the C ABI: every tensor the chain writes."""
    _, fa, _, _ = mods
    from butd_detr_amd import _hiplib
    lib = _hiplib.load()
    torch.manual_seed(rows + force)
    E, Fh = 288, 256
    dev = "cuda"
    att = torch.randn(rows, E, device=dev)
    x = torch.randn(rows, E, device=dev)
    pos = torch.randn(rows, E, device=dev)
    w_o, b_o = torch.randn(E, E, device=dev) / 17, torch.randn(E, device=dev) * 0.1
    w_q, b_q = torch.randn(E, E, device=dev) / 17, torch.randn(E, device=dev) * 0.1
    w1, b1 = torch.randn(Fh, E, device=dev) / 17, torch.randn(Fh, device=dev) * 0.1
    w2, b2 = torch.randn(E, Fh, device=dev) / 16, torch.randn(E, device=dev) * 0.1
    n1, n2 = _rand_ln(E), _rand_ln(E)
    proj, y, ypos, q = (torch.empty(rows, E, device=dev) for _ in range(4))
    h, o2, y2 = torch.empty(rows, Fh, device=dev), torch.empty(rows, E, device=dev), torch.empty(rows, E, device=dev)
    mean, rstd, mean2, rstd2 = (torch.empty(rows, device=dev) for _ in range(4))
    assert lib.butd_panel_set_rows(force) == 0
    try:
        fa._panel(rows, att, E, [
            fa._stage(w_o, b_o, E, E, 0, 1, pre=proj, ln=(n1.weight, n1.bias, n1.eps, mean, rstd), res=x, out=y, pos=pos,
                      out_pos=ypos, pos_buf=2),
            fa._stage(w_q, b_q, E, E, 2, 0, scale=1 / 6.0, out=q),
            fa._stage(w1, b1, Fh, E, 1, 3, relu=True, out=h),
            fa._stage(w2, b2, E, Fh, 3, 0, pre=o2, ln=(n2.weight, n2.bias, n2.eps, mean2, rstd2), res_buf=1, out=y2)],
            4, att)
    finally:
        lib.butd_panel_set_rows(0)
    with torch.no_grad():
        a64 = att.double()
        proj_r = a64 @ w_o.double().T + b_o.double()
        s = x.double() + proj_r
        mu = s.mean(-1, keepdim=True)
        var = s.var(-1, unbiased=False, keepdim=True)
        y_r = (s - mu) / torch.sqrt(var + n1.eps) * n1.weight.double() + n1.bias.double()
        q_r = ((y_r + pos.double()) @ w_q.double().T + b_q.double()) / 6.0
        h_r = torch.relu(y_r @ w1.double().T + b1.double())
        o_r = h_r @ w2.double().T + b2.double()
        s2 = y_r + o_r
        mu2 = s2.mean(-1, keepdim=True)
        y2_r = (s2 - mu2) / torch.sqrt(s2.var(-1, unbiased=False, keepdim=True) + n2.eps) * n2.weight.double() + n2.bias.double()
    for got, want, name in ((proj, proj_r, "proj"), (y, y_r, "y"), (ypos, y_r + pos.double(), "y_pos"), (q, q_r, "q"),
                            (h, h_r, "h"), (o2, o_r, "o"), (y2, y2_r, "y2"), (mean, mu[:, 0], "mean"),
                            (rstd, 1 / torch.sqrt(var[:, 0] + n1.eps), "rstd"), (mean2, mu2[:, 0], "mean2")):
        _close(got, want, 2e-5 if name != "h" else 5e-5, name)     # fp32 MFMA is exact fp32: far inside north_star's 1e-3


def _decoder_like(mods, B, Q, Lk, p):
    ab, fa, MHA, _ffn = mods
    E, H = 288, 8
    torch.manual_seed(B * Q + Lk)
    mk = lambda: MHA(E, H, dropout=p).cuda()
    blocks = [mk(), mk(), mk()]
    for a in blocks:
        with torch.no_grad():
            a.in_proj_bias.uniform_(-0.1, 0.1)
            a.out_proj.bias.uniform_(-0.1, 0.1)
    norms = [_rand_ln(E) for _ in range(4)]
    drops = [torch.nn.Dropout(p) for _ in range(3)]
    ffn = _ffn(E, 256, p).cuda()
    x = torch.randn(B, Q, E, device="cuda", requires_grad=True)
    pos = torch.randn(B, Q, E, device="cuda", requires_grad=True)
    mem = torch.randn(B, Lk, E, device="cuda", requires_grad=True)
    mask = torch.zeros(B, Lk, dtype=torch.bool, device="cuda")
    mask[0, Lk - 3:] = True
    mods_all = torch.nn.ModuleList(blocks + norms + [ffn])

    def run():
        """self-attention -> cross-attention -> cross-attention + FFN, chained the way BiDecoderLayer chains them"""
        first = lambda em: em[0] if em else None
        y, yp, em = ab.block(blocks[0], drops[0], norms[0], x=x, pos=pos, next_pos=pos,
                             emit=[ab.q_projection(blocks[1], True)])
        y, yp, em = ab.block(blocks[1], drops[1], norms[1], x=y, pos=pos, xq_pre=yp, q_pre=first(em), memory=mem,
                             key_padding_mask=mask, next_pos=pos, emit=[ab.q_projection(blocks[2], True)])
        y = ab.block(blocks[2], drops[2], norms[2], x=y, pos=pos, xq_pre=yp, q_pre=first(em), memory=mem,
                     key_padding_mask=mask, ffn=(ffn, norms[3]))[0]
        return y
    return run, mods_all, (x, pos, mem)


@pytest.mark.parametrize("B,Q,Lk", [(2, 64, 40), (8, 256, 80), (3, 50, 132)])
def test_chained_blocks_match_torch(mods, B, Q, Lk):
    ab, fa, _, _ = mods
    run, mods_all, inputs = _decoder_like(mods, B, Q, Lk, 0.0)
    mods_all.train()
    probe = torch.randn(B, Q, 288, device="cuda")
    outs = {}
    for backend in ("torch", "hip"):
        ab.set_backend(backend)
        try:
            for t in list(inputs) + list(mods_all.parameters()):
                t.grad = None
            y = run()
            (y * probe).sum().backward()
            outs[backend] = (y.detach().clone(), [t.grad.clone() for t in list(inputs) + list(mods_all.parameters())])
        finally:
            ab.set_backend("torch")
    _close(outs["hip"][0], outs["torch"][0], 1e-3, "y")
    for i, (gh, gt) in enumerate(zip(outs["hip"][1], outs["torch"][1])):
        _close(gh, gt, 2e-3, f"grad {i}")


@pytest.mark.parametrize("B,Q,Lk", [(8, 256, 80), (8, 1024, 80)])
def test_chain_equals_unchained_path_under_dropout(mods, B, Q, Lk):
    """Same step counter, same sites -> the chain draws the masks the per-operator kernels draw: outputs and gradients
    agree to rounding (the LayerNorm sums are taken in a different order), with dropout ON everywhere."""
    ab, fa, _, _ = mods
    run, mods_all, inputs = _decoder_like(mods, B, Q, Lk, 0.1)
    mods_all.train()
    probe = torch.randn(B, Q, 288, device="cuda")
    ctr = fa.rng_counter(torch.device("cuda", 0))
    outs = {}
    ab.set_backend("hip")
    try:
        for chained in (False, True):
            prev = fa.set_panel_chain(chained)
            try:
                ctr.fill_(4242)
                fa._site[0] = 0
                for t in list(inputs) + list(mods_all.parameters()):
                    t.grad = None
                y = run()
                (y * probe).sum().backward()
                outs[chained] = (y.detach().clone(), [t.grad.clone() for t in list(inputs) + list(mods_all.parameters())])
            finally:
                fa.set_panel_chain(prev)
    finally:
        ab.set_backend("torch")
    # (sites are numbered by block() before either path launches anything: the masks are the same)
    _close(outs[True][0], outs[False][0], 1e-4, "y", frac=1e-2)
    for i, (gc, gu) in enumerate(zip(outs[True][1], outs[False][1])):
        _close(gc, gu, 2e-4, f"grad {i}", frac=1e-2)


def test_ffn_block_chain_vs_unchained_same_masks(mods):
    """The stand-alone FFN block numbers its sites identically on both paths: with dropout on, chain == unchained."""
    ab, fa, _, _ffn = mods
    torch.manual_seed(5)
    E = 288
    ffn = _ffn(E, 256, 0.1).cuda().train()
    norm = _rand_ln(E)
    x = torch.randn(8, 256, E, device="cuda", requires_grad=True)
    probe = torch.randn_like(x)
    ctr = fa.rng_counter(torch.device("cuda", 0))
    outs = {}
    for chained in (False, True):
        prev = fa.set_panel_chain(chained)
        try:
            ctr.fill_(99)
            fa._site[0] = 10
            for t in [x] + list(ffn.parameters()) + list(norm.parameters()):
                t.grad = None
            y = fa.ffn_block(ffn, norm, x)
            (y * probe).sum().backward()
            outs[chained] = [y.detach().clone()] + [t.grad.clone() for t in [x] + list(ffn.parameters()) + list(norm.parameters())]
        finally:
            fa.set_panel_chain(prev)
    for a, b in zip(outs[True], outs[False]):
        _close(a, b, 1e-5)
