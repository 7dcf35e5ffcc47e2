"""-m gpu: fused gfx950 attention / FFN blocks vs the plain-torch maths (the arithmetic of
torch.nn.functional.multi_head_attention_forward), forward and backward, tolerance 1e-3 fp32
(north_star).  Dropout is checked for determinism, keep-rate and fwd/bwd mask consistency."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    assert torch.cuda.is_available()
    from butd_detr_amd import attention_blocks as ab
    from butd_detr_amd import fused_attention as fa
    from butd_detr_amd.encoder_decoder_layers import MultiheadAttention, _ffn
    return ab, fa, MultiheadAttention, _ffn


def _close(a, b, tol=1e-3):
    a, b = a.detach().float().cpu().numpy(), b.detach().float().cpu().numpy()
    scale = max(np.abs(b).max(), 1e-6)
    np.testing.assert_allclose(a / scale, b / scale, rtol=0, atol=tol)


@pytest.mark.parametrize("B,Lq,Lk,masked", [(2, 64, 64, False), (2, 80, 1024, False), (3, 1024, 80, True),
                                            (2, 256, 132, True), (1, 17, 5, True), (2, 100, 100, True)])
def test_attention_block_matches_torch(mods, B, Lq, Lk, masked):
    ab, fa, MHA, _ = mods
    torch.manual_seed(Lq + Lk)
    E, H = 288, 8
    attn = MHA(E, H, dropout=0.1).cuda().eval()
    with torch.no_grad():
        attn.in_proj_bias.uniform_(-0.1, 0.1)
        attn.out_proj.bias.uniform_(-0.1, 0.1)
    norm = torch.nn.LayerNorm(E).cuda()
    with torch.no_grad():
        norm.weight.uniform_(0.8, 1.2)
        norm.bias.uniform_(-0.1, 0.1)
    drop = torch.nn.Dropout(0.1).eval()
    self_attn = Lq == Lk
    res = torch.randn(B, Lq, E, device="cuda", requires_grad=True)
    xq = torch.randn(B, Lq, E, device="cuda", requires_grad=True)
    xk = xq if self_attn else torch.randn(B, Lk, E, device="cuda", requires_grad=True)
    xv = torch.randn(B, Lk, E, device="cuda", requires_grad=True)
    mask = None
    if masked:
        mask = torch.zeros(B, Lk, dtype=torch.bool, device="cuda")
        for b in range(B):
            mask[b, Lk - 1 - b:] = True
    probe = torch.randn(B, Lq, E, device="cuda")
    params = list(attn.parameters()) + list(norm.parameters())

    def run(fn):
        for t in [res, xq, xk, xv] + params:
            t.grad = None
        y = fn(attn, drop, norm, residual=res, query=xq, key=xk, value=xv, key_padding_mask=mask)
        (y * probe).sum().backward()
        return y, [t.grad.clone() for t in ([res, xq, xv] + ([] if self_attn else [xk]) + params)]

    y_ref, g_ref = run(lambda a, d, n, **kw: n(kw["residual"] + d(ab._mha_torch(a, kw["query"], kw["key"], kw["value"], kw["key_padding_mask"]))))
    y_hip, g_hip = run(lambda a, d, n, **kw: fa.attention_block(a, d, n, kw["residual"], kw["query"], kw["key"], kw["value"], kw["key_padding_mask"]))
    _close(y_hip, y_ref)
    for gh, gr in zip(g_hip, g_ref):
        _close(gh, gr, 2e-3)


@pytest.mark.parametrize("pattern", ["self_pos", "self", "cross_pos", "cross"])
@pytest.mark.parametrize("B,Lq,Lk,masked", [(2, 64, 64, False), (3, 256, 132, True), (2, 80, 1024, False),
                                            (1, 17, 5, True)])
def test_structured_block_matches_torch(mods, pattern, B, Lq, Lk, masked):
    """attention_blocks.block (x, pos, memory): the fused backward returns SUMMED gradients (packed
    dq|dk|dv, c_add / c2 epilogues); they must equal what autograd accumulates for the torch maths."""
    ab, fa, MHA, _ = mods
    torch.manual_seed(Lq * 7 + Lk)
    E, H = 288, 8
    is_self = pattern.startswith("self")
    if is_self:
        Lk = Lq
    attn = MHA(E, H, dropout=0.1).cuda().eval()
    norm = torch.nn.LayerNorm(E).cuda()
    with torch.no_grad():
        attn.in_proj_bias.uniform_(-0.1, 0.1)
        attn.out_proj.bias.uniform_(-0.1, 0.1)
        norm.weight.uniform_(0.8, 1.2)
        norm.bias.uniform_(-0.1, 0.1)
    drop = torch.nn.Dropout(0.1).eval()
    x = torch.randn(B, Lq, E, device="cuda", requires_grad=True)
    pos = torch.randn(B, Lq, E, device="cuda", requires_grad=True) if pattern.endswith("pos") else None
    mem = None if is_self else torch.randn(B, Lk, E, device="cuda", requires_grad=True)
    mask = None
    if masked:
        mask = torch.zeros(B, Lk, dtype=torch.bool, device="cuda")
        for b in range(B):
            mask[b, Lk - 1 - b:] = True
    probe = torch.randn(B, Lq, E, device="cuda")
    leaves = [t for t in (x, pos, mem) if t is not None]
    params = list(attn.parameters()) + list(norm.parameters())

    def run(backend):
        ab.set_backend(backend)
        for t in leaves + params:
            t.grad = None
        # consume x twice more outside the block, as the layers do (other gradient contributions)
        y = ab.block(attn, drop, norm, x=x, pos=pos, memory=mem, key_padding_mask=mask)
        ((y * probe).sum() + (x * x).sum() * 0.01).backward()
        return y, [t.grad.clone() for t in leaves + params]

    try:
        y_ref, g_ref = run("torch")
        y_hip, g_hip = run("hip")
    finally:
        ab.set_backend("torch")
    _close(y_hip, y_ref)
    for gh, gr in zip(g_hip, g_ref):
        _close(gh, gr, 2e-3)


def test_fully_masked_row_is_nan_like_torch(mods):
    ab, fa, MHA, _ = mods
    attn = MHA(288, 8, dropout=0.0).cuda().eval()
    norm = torch.nn.LayerNorm(288).cuda()
    x = torch.randn(2, 16, 288, device="cuda")
    kv = torch.randn(2, 10, 288, device="cuda")
    mask = torch.zeros(2, 10, dtype=torch.bool, device="cuda")
    mask[1] = True
    y = fa.attention_block(attn, None, norm, x, x, kv, kv, mask)
    ref = norm(x + ab._mha_torch(attn, x, kv, kv, mask))
    assert torch.isnan(y[1]).all() and torch.isnan(ref[1]).all()
    _close(y[0], ref[0])


@pytest.mark.parametrize("B,L", [(2, 80), (8, 1024), (1, 7)])
def test_ffn_block_matches_torch(mods, B, L):
    ab, fa, _, make_ffn = mods
    torch.manual_seed(L)
    E = 288
    ffn = make_ffn(E, 256, 0.1).cuda().eval()
    norm = torch.nn.LayerNorm(E).cuda()
    x = torch.randn(B, L, E, device="cuda", requires_grad=True)
    probe = torch.randn(B, L, E, device="cuda")
    params = list(ffn.parameters()) + list(norm.parameters())

    def run(fn):
        for t in [x] + params:
            t.grad = None
        y = fn(x)
        (y * probe).sum().backward()
        return y, [t.grad.clone() for t in [x] + params]

    y_ref, g_ref = run(lambda t: norm(t + ffn(t)))
    y_hip, g_hip = run(lambda t: fa.ffn_block(ffn, norm, t))
    _close(y_hip, y_ref)
    for gh, gr in zip(g_hip, g_ref):
        _close(gh, gr, 2e-3)


def test_dropout_is_deterministic_per_step_and_consistent_between_fwd_and_bwd(mods):
    ab, fa, MHA, make_ffn = mods
    torch.manual_seed(0)
    E = 288
    ffn = make_ffn(E, 256, 0.1).cuda().train()
    norm = torch.nn.LayerNorm(E).cuda()
    x = torch.randn(4, 256, E, device="cuda", requires_grad=True)
    dev = x.device

    def f(t):
        fa._site[0] = 100            # same sites -> same masks within a step
        return fa.ffn_block(ffn, norm, t)

    fa.new_step(dev)
    y1 = f(x)
    y2 = f(x)
    assert torch.equal(y1, y2)
    # directional derivative vs autograd: masks must be identical in forward and backward
    v = torch.randn_like(x)
    probe = torch.randn_like(x)
    (y1 * probe).sum().backward()
    analytic = (x.grad * v).sum().item()
    eps = 1e-2
    with torch.no_grad():
        num = (((f(x + eps * v) - f(x - eps * v)) / (2 * eps)) * probe).sum().item()
    assert abs(num - analytic) <= 3e-2 * max(abs(analytic), 1.0)
    fa.new_step(dev)
    y3 = f(x)
    assert not torch.equal(y1, y3)   # a new step draws a new mask


def test_attention_dropout_keep_rate(mods):
    ab, fa, MHA, _ = mods
    attn = MHA(288, 8, dropout=0.3).cuda().train()
    norm = torch.nn.LayerNorm(288).cuda()
    with torch.no_grad():            # V projection = identity-ish probe: out = mean of kept probs
        attn.in_proj_weight.zero_()
        attn.in_proj_bias.zero_()
        attn.in_proj_bias[2 * 288:] = 1.0      # every value vector = ones -> context = sum of kept p/(1-p)
        attn.out_proj.weight.copy_(torch.eye(288))
        attn.out_proj.bias.zero_()
    x = torch.randn(2, 512, 288, device="cuda")
    fa.new_step(x.device)
    q = torch.zeros_like(x)
    # uniform attention (q = 0): context = (1/Lk) * sum_k keep_k/(1-p); expectation 1, variance small
    out = fa._AttentionBlock.apply(torch.zeros_like(x), q, x, x, None, attn.in_proj_weight, attn.in_proj_bias,
                                   attn.out_proj.weight, attn.out_proj.bias, torch.ones(288, device="cuda"),
                                   torch.zeros(288, device="cuda"), 8, 0.0, 0.3, 0.0, 7, 8)
    # LayerNorm of a constant row is 0; instead read the pre-norm via a second call without norm effect
    # -> check through the saved context statistics: rerun the core directly
    import ctypes
    from butd_detr_amd import _hiplib
    lib = _hiplib.load()
    B, L, E, H, D = 2, 512, 288, 8, 36
    qq = torch.zeros(B, L, E, device="cuda")
    kk = torch.zeros(B, L, E, device="cuda")
    vv = torch.ones(B, L, E, device="cuda")
    o = torch.empty(B, L, E, device="cuda")
    lse = torch.empty(B, H, L, device="cuda")
    err = lib.butd_attention_fwd(B, H, L, L, D, qq.data_ptr(), kk.data_ptr(), vv.data_ptr(), None, o.data_ptr(),
                                 lse.data_ptr(), 0.3, 11, fa.rng_counter(x.device).data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
    assert err == 0
    torch.cuda.synchronize()
    assert abs(o.mean().item() - 1.0) < 0.01
    assert 0.02 < o.std().item() < 0.06          # sqrt(p/(1-p)/Lk) = 0.029


@pytest.mark.parametrize("pattern,B,Lq,Lk,masked", [("self_pos", 8, 1024, 1024, False), ("cross_pos", 8, 256, 1024, False),
                                                    ("cross_pos", 4, 1024, 132, True), ("cross", 2, 80, 1024, False),
                                                    ("self", 3, 100, 100, True)])
def test_bf16_matrix_steps_track_fp32(mods, pattern, B, Lq, Lk, masked):
    """BASELINE configs[3] ("bf16 attention"): butd_attention_fwd_bf16 / _bwd_bf16 (and the bf16 grouped products
    around them) against the fp32 kernels on the same block: the operands of every matrix step are rounded to
    bf16 (2^-8), accumulation and softmax statistics are fp32 -- outputs within 2e-2 of the tensor scale,
    gradients within 5e-2, on the call-site shapes incl. the benchmarked 1024 x 1024."""
    ab, fa, MHA, _ = mods
    torch.manual_seed(Lq + 3 * Lk)
    E, H = 288, 8
    is_self = pattern.startswith("self")
    attn = MHA(E, H, dropout=0.0).cuda().train()
    norm = torch.nn.LayerNorm(E).cuda()
    with torch.no_grad():
        attn.in_proj_bias.uniform_(-0.1, 0.1)
        norm.weight.uniform_(0.8, 1.2)
    drop = torch.nn.Dropout(0.0)
    x = torch.randn(B, Lq, E, device="cuda", requires_grad=True)
    pos = torch.randn(B, Lq, E, device="cuda", requires_grad=True) if pattern.endswith("pos") else None
    mem = None if is_self else torch.randn(B, Lk, E, device="cuda", requires_grad=True)
    mask = None
    if masked:
        mask = torch.zeros(B, Lk, dtype=torch.bool, device="cuda")
        for b in range(B):
            mask[b, Lk - 1 - 3 * b:] = True
    probe = torch.randn(B, Lq, E, device="cuda")
    leaves = [t for t in (x, pos, mem) if t is not None]
    params = list(attn.parameters()) + list(norm.parameters())

    def run(dtype):
        fa.set_compute_dtype(dtype)
        for t in leaves + params:
            t.grad = None
        y = ab.block(attn, drop, norm, x=x, pos=pos, memory=mem, key_padding_mask=mask)
        (y * probe).sum().backward()
        return y, [t.grad.clone() for t in leaves + params]

    try:
        ab.set_backend("hip")
        y32, g32 = run("f32")
        y16, g16 = run("bf16")
    finally:
        fa.set_compute_dtype("f32")
        ab.set_backend("torch")
    assert not torch.equal(y16, y32)
    _close(y16, y32, 2e-2)
    for a, b in zip(g16, g32):
        _close(a, b, 5e-2)


def test_bf16_attention_core_alone(mods):
    """The core by itself (no projections): bf16 entry points vs the fp32 ones on random q, k, v -- forward within
    1e-2 of the output scale, dq / dk / dv within 3e-2 (dropout 0.1 on: both draw the same counter-hash masks)."""
    from butd_detr_amd import _hiplib
    ab, fa, _, _ = mods
    lib = _hiplib.load()
    B, H, D, Lq, Lk = 4, 8, 36, 512, 320
    E = H * D
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v, do = (torch.randn(B, L, E, device="cuda", generator=g) * 0.5 for L in (Lq, Lk, Lk, Lq))
    ctr = fa.rng_counter(q.device).data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for name, fwd, bwd in (("f32", lib.butd_attention_fwd, lib.butd_attention_bwd),
                           ("bf16", lib.butd_attention_fwd_bf16, lib.butd_attention_bwd_bf16)):
        out, dq = torch.empty_like(q), torch.empty_like(q)
        dk, dv = torch.empty_like(k), torch.empty_like(v)
        lse, delta = torch.empty(B, H, Lq, device="cuda"), torch.empty(B, H, Lq, device="cuda")
        assert fwd(B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), None, out.data_ptr(), lse.data_ptr(),
                   0.1, 5, ctr, st) == 0
        assert bwd(B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), None, out.data_ptr(), do.data_ptr(),
                   lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), 0, 0, 1.0, 0.1, 5,
                   ctr, st) == 0
        torch.cuda.synchronize()
        res[name] = (out, dq, dk, dv)
    _close(res["bf16"][0], res["f32"][0], 1e-2)
    for i in (1, 2, 3):
        _close(res["bf16"][i], res["f32"][i], 3e-2)


@pytest.mark.parametrize("Lq,Lk,masked", [(512, 320, False), (256, 132, True), (100, 77, True)])
def test_split_bf16_forward_core_tracks_fp32_and_float64(mods, Lq, Lk, masked):
    """Round 6's microbenchmark kernel (butd_attention_fwd_split_bf16: hi + lo bf16 operands, three products per term):
    against a float64 evaluation of the same attention within 5e-5 of the output scale (the fp32 kernel: ~1e-6, the bf16
    kernel: ~8e-3), ragged key tiles and a key-padding mask included; the log-sum-exp within 1e-5.  Dropout draws the same
    counter-hash mask as the fp32 kernel."""
    from butd_detr_amd import _hiplib
    ab, fa, _, _ = mods
    lib = _hiplib.load()
    B, H, D = 3, 8, 36
    E = H * D
    g = torch.Generator(device="cuda").manual_seed(Lq)
    q = torch.randn(B, Lq, E, device="cuda", generator=g) / 6.0
    k, v = (torch.randn(B, Lk, E, device="cuda", generator=g) for _ in range(2))
    mask = None
    if masked:
        mask = torch.zeros(B, Lk, dtype=torch.bool, device="cuda")
        mask[0, Lk // 2:] = True
        mask[2, 5:9] = True
    mptr = None if mask is None else mask.data_ptr()
    ctr = fa.rng_counter(q.device).data_ptr()
    st = torch.cuda.current_stream().cuda_stream

    def run(fn, p):
        out, lse = torch.empty_like(q), torch.empty(B, H, Lq, device="cuda")
        assert fn(B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), mptr, out.data_ptr(), lse.data_ptr(), p, 9, ctr, st) == 0
        torch.cuda.synchronize()
        return out, lse
    qd, kd, vd = (t.double().view(B, -1, H, D).transpose(1, 2) for t in (q, k, v))
    sc = qd @ kd.transpose(-1, -2)
    if mask is not None:
        sc = sc.masked_fill(mask[:, None, None, :], float("-inf"))
    truth = (torch.softmax(sc, -1) @ vd).transpose(1, 2).reshape(B, Lq, E)
    out, lse = run(lib.butd_attention_fwd_split_bf16, 0.0)
    scale = float(truth.abs().max())
    assert float((out.double() - truth).abs().max()) / scale < 5e-5
    assert float((lse.double() - torch.logsumexp(sc, -1)).abs().max()) < 1e-5 * max(1.0, float(torch.logsumexp(sc, -1).abs().max()))
    o32, _ = run(lib.butd_attention_fwd, 0.1)
    osp, _ = run(lib.butd_attention_fwd_split_bf16, 0.1)
    assert float((osp - o32).abs().max()) / scale < 1e-4                    # same keep mask, same scaling by 1 / (1 - p)


@pytest.mark.parametrize("B,Lq,Lk,masked", [(8, 1024, 1024, False), (8, 256, 1024, False), (8, 80, 1024, True), (4, 200, 1500, True),
                                            (8, 256, 256, False), (8, 256, 132, True), (8, 1024, 132, True), (8, 1024, 80, True),
                                            (8, 80, 80, True), (3, 77, 300, True)])
def test_bf16_long_key_backward_in_one_pass(mods, B, Lq, Lk, masked):
    """butd_attention_bwd_long_keys_bf16 (one pass, bf16 images, K = 32 matrix instruction) against the two bf16 kernels on
    the same saved forward (the dropout scale rides on V here and on dO in the dQ kernel, so other operand roundings: within
    1.5e-2 of the scale of each other) and against the fp32 backward (3e-2, the bound of the bf16 entry points, and no
    further from fp32 than the two kernels are); packed gradient rows; bit-identical between two runs."""
    from butd_detr_amd import _hiplib
    ab, fa, _, _ = mods
    lib = _hiplib.load()
    H, D = 8, 36
    E = H * D
    g = torch.Generator(device="cuda").manual_seed(Lq + Lk)
    q, k, v, do = (torch.randn(B, L, E, device="cuda", generator=g) * 0.5 for L in (Lq, Lk, Lk, Lq))
    mask = None
    if masked:
        mask = torch.zeros(B, Lk, dtype=torch.uint8, device="cuda")
        for b in range(B):
            mask[b, Lk - 1 - min(37 * b, Lk // 2):] = 1
    mp = mask.data_ptr() if mask is not None else None
    ctr = fa.rng_counter(q.device).data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    out, lse = torch.empty_like(q), torch.empty(B, H, Lq, device="cuda")
    assert lib.butd_attention_fwd_bf16(B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), mp, out.data_ptr(),
                                       lse.data_ptr(), 0.1, 5, ctr, st) == 0
    ldq, ldkv = E + 4, 2 * E
    assert lib.butd_attention_bwd_long_keys_bf16_scratch(1, 1, 64, 64, D, ldq) == -1      # would leave the part idle
    need = int(lib.butd_attention_bwd_long_keys_bf16_scratch(B, H, Lq, Lk, D, ldq))
    if need < 0:         # (few query tiles over >= 4 chunks, e.g. 256 x 256, stay on the two bf16 kernels: measured equal)
        assert Lq <= 256 and Lk > 192         # -- exercised here through the tuning hook
        lib.butd_attention_bwd_long_keys_set_chunk(64, 1)
        need = int(lib.butd_attention_bwd_long_keys_bf16_scratch(B, H, Lq, Lk, D, ldq))
    assert need == int(lib.butd_attention_bwd_long_keys_scratch(B, H, Lq, Lk, D, ldq)) and need >= 0     # the fp32 plan
    res = {}
    for name in ("f32", "two", "one", "one again"):
        dqb = torch.full((B, Lq, ldq), 7.0, device="cuda")
        G = torch.full((B, Lk, ldkv), 7.0, device="cuda")
        delta = torch.empty(B, H, Lq, device="cuda")
        args = (B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), mp, out.data_ptr(), do.data_ptr(), lse.data_ptr())
        outs = (dqb.data_ptr(), G.data_ptr(), G.data_ptr() + 4 * E, ldq, ldkv, 0.5, 0.1, 5, ctr)
        if name == "f32":
            assert lib.butd_attention_bwd(*args, delta.data_ptr(), *outs, st) == 0
        elif name == "two":
            assert lib.butd_attention_bwd_bf16(*args, delta.data_ptr(), *outs, st) == 0
        else:
            ws = torch.full((max(need, 1),), float("nan"), device="cuda")
            assert lib.butd_attention_bwd_long_keys_bf16(*args, *outs, ws.data_ptr(), need, st) == 0
        torch.cuda.synchronize()
        res[name] = (dqb[:, :, :E].clone(), G[:, :, :E].clone(), G[:, :, E:].clone(), dqb[:, :, E:].clone())
    lib.butd_attention_bwd_long_keys_set_chunk(0, 0)
    assert float(res["one"][3].min()) == 7.0 and float(res["one"][3].max()) == 7.0
    err = lambda a, b: float((a - b).abs().max() / b.abs().max())
    for i, name in enumerate(("dq", "dk", "dv")):
        assert torch.isfinite(res["one"][i]).all()
        e_one, e_two = err(res["one"][i], res["f32"][i]), err(res["two"][i], res["f32"][i])
        print(f"{name}: one pass vs fp32 {e_one:.2e}, two kernels vs fp32 {e_two:.2e}, one vs two {err(res['one'][i], res['two'][i]):.2e}")
        assert e_one <= 3e-2 and e_one <= 1.5 * e_two + 1e-3, (name, e_one, e_two)      # no further from fp32 than the two kernels
        _close(res["one"][i], res["two"][i], 1.5e-2)
        assert torch.equal(res["one"][i], res["one again"][i])


def test_chain_of_blocks_with_the_position_sum_from_the_layernorm_kernel(mods):
    """``next_pos`` / ``xq_pre`` (butd_add_dropout_layernorm_fwd_pos): a block writes `output + pos` for the next
    block, which uses it in place of its own `x + pos` -- values and every gradient (x, pos, memory, parameters) equal
    the chain that adds in each block, and the torch maths (the decoder layer's four blocks,
    encoder_decoder_layers.py:356-404)."""
    ab, fa, MHA, _ = mods
    torch.manual_seed(5)
    E, H, B, Lq, Lk = 288, 8, 3, 256, 80
    blocks = []
    for _ in range(3):
        attn = MHA(E, H, dropout=0.0).cuda().eval()
        norm = torch.nn.LayerNorm(E).cuda()
        with torch.no_grad():
            norm.weight.uniform_(0.8, 1.2)
            norm.bias.uniform_(-0.1, 0.1)
        blocks.append((attn, norm))
    drop = torch.nn.Dropout(0.0).eval()
    x = torch.randn(B, Lq, E, device="cuda", requires_grad=True)
    pos = torch.randn(B, Lq, E, device="cuda", requires_grad=True)
    mem = torch.randn(B, Lk, E, device="cuda", requires_grad=True)
    probe = torch.randn(B, Lq, E, device="cuda")
    leaves = [x, pos, mem] + [p for a, n in blocks for p in list(a.parameters()) + list(n.parameters())]

    def run(backend, chained):
        ab.set_backend(backend)
        for t in leaves:
            t.grad = None
        q, qp = x, None
        for i, (attn, norm) in enumerate(blocks):
            memory = None if i == 0 else mem
            last = i == len(blocks) - 1
            if chained:
                r = ab.block(attn, drop, norm, x=q, pos=pos, memory=memory, xq_pre=qp,
                             next_pos=None if last else pos)
                q, qp = (r[0], r[1]) if isinstance(r, tuple) else (r, None)
            else:
                q = ab.block(attn, drop, norm, x=q, pos=pos, memory=memory)
        (q * probe).sum().backward()
        return q, [t.grad.clone() for t in leaves]

    try:
        y_ref, g_ref = run("torch", False)
        y_add, g_add = run("hip", False)
        y_ch, g_ch = run("hip", True)
    finally:
        ab.set_backend("torch")
    assert torch.equal(y_ch, y_add)                  # same arithmetic: x + pos is one rounding either way
    for a, b in zip(g_ch, g_add):                    # (split-K weight gradients add their slices in arrival order)
        _close(a, b, 1e-5)
    _close(y_ch, y_ref)
    for a, b in zip(g_ch, g_ref):
        _close(a, b, 2e-3)


@pytest.mark.parametrize("lead", [(7, 8, 256), (8, 80), (3, 5)])
def test_linear_relu_chain_matches_the_stock_modules(mods, lead):
    """fused_attention.linear_relu_chain: the contrastive-alignment projection MLPs (bdetr.py:104-121) on the grouped
    GEMM -- ReLU in the forward epilogues, the ReLU gate in the epilogue of the product that creates each hidden
    gradient (c_gate) -- values and all gradients vs nn.Sequential."""
    ab, fa, _, _ = mods
    torch.manual_seed(sum(lead))
    seq = torch.nn.Sequential(torch.nn.Linear(288, 288), torch.nn.ReLU(), torch.nn.Linear(288, 288), torch.nn.ReLU(),
                              torch.nn.Linear(288, 64)).cuda()
    x = torch.randn(*lead, 288, device="cuda", requires_grad=True)
    probe = torch.randn(*lead, 64, device="cuda")
    leaves = [x] + list(seq.parameters())

    def run(fn):
        for t in leaves:
            t.grad = None
        y = fn(x)
        (torch.nn.functional.normalize(y, p=2, dim=-1) * probe).sum().backward()
        return y, [t.grad.clone() for t in leaves]

    y_ref, g_ref = run(seq)
    y_hip, g_hip = run(lambda t: fa.linear_relu_chain(seq, t))
    _close(y_hip, y_ref)
    for a, b in zip(g_hip, g_ref):
        _close(a, b, 2e-3)
    # anything else than Linear / ReLU alternation goes to the modules themselves
    odd = torch.nn.Sequential(torch.nn.Linear(288, 288), torch.nn.Tanh(), torch.nn.Linear(288, 64)).cuda()
    assert torch.equal(fa.linear_relu_chain(odd, x), odd(x))


@pytest.mark.parametrize("B,Lq,Lk,masked,p", [(8, 256, 80, True, 0.1), (8, 256, 132, True, 0.1), (8, 1024, 80, True, 0.1),
                                              (4, 1024, 132, False, 0.0), (8, 80, 80, True, 0.1), (2, 17, 5, True, 0.0),
                                              (3, 100, 144, False, 0.1), (2, 200, 33, True, 0.1)])
def test_short_key_backward_equals_two_kernel_backward(mods, B, Lq, Lk, masked, p):
    """butd_attention_bwd_short_keys (one kernel, every score tile computed once, dK / dV accumulated with atomics)
    against butd_attention_bwd (dQ kernel + dK/dV kernel) on the same saved forward: same dropout masks (same hash),
    equal to fp32 summation order (1e-5 of the scale); packed / strided gradient rows as the structured block
    passes them."""
    _, fa, _, _ = mods
    from butd_detr_amd import _hiplib
    lib = _hiplib.load()
    torch.manual_seed(Lq * 3 + Lk)
    H, D = 8, 36
    E = H * D
    dev = "cuda"
    q, do = torch.randn(B, Lq, E, device=dev) / 6, torch.randn(B, Lq, E, device=dev)
    k, v = torch.randn(B, Lk, E, device=dev), torch.randn(B, Lk, E, device=dev)
    mask = None
    if masked:
        mask = torch.zeros(B, Lk, dtype=torch.uint8, device=dev)
        for b in range(B):
            mask[b, Lk - 1 - (b % min(Lk - 1, 7)):] = 1
    out, lse = torch.empty_like(q), torch.empty(B, H, Lq, device=dev)
    ctr = fa.rng_counter(torch.device("cuda", 0))
    ctr.fill_(77)
    st = torch.cuda.current_stream().cuda_stream
    mp = mask.data_ptr() if mask is not None else None
    assert lib.butd_attention_fwd(B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), mp, out.data_ptr(),
                                  lse.data_ptr(), p, 5, ctr.data_ptr(), st) == 0
    res = {}
    for name, fn in (("two", lib.butd_attention_bwd), ("one", lib.butd_attention_bwd_short_keys)):
        dq = torch.full((B, Lq, E), float("nan"), device=dev)
        G = torch.zeros(B, Lk, 2 * E + 4, device=dev)        # dk | dv side by side, rows wider than 2E
        delta = torch.empty(B, H, Lq, device=dev)
        assert fn(B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), mp, out.data_ptr(), do.data_ptr(),
                  lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), G.data_ptr(), G.data_ptr() + 4 * E, E, 2 * E + 4, 0.5,
                  p, 5, ctr.data_ptr(), st) == 0
        torch.cuda.synchronize()
        res[name] = (dq, G[:, :, :E].clone(), G[:, :, E:2 * E].clone(), delta, G[:, :, 2 * E:].clone())
    for a, bq, name in zip(res["one"], res["two"], ("dq", "dk", "dv", "delta", "padding")):
        assert torch.isfinite(a).all(), name
        _close(a, bq, 1e-5 if name != "padding" else 1e-12)


@pytest.mark.parametrize("B,Lq,Lk,masked,p,force", [
    (8, 1024, 1024, False, 0.1, None), (8, 256, 1024, False, 0.1, None), (8, 80, 1024, True, 0.1, None),
    (8, 256, 256, False, 0.1, None), (8, 256, 132, True, 0.1, None), (8, 80, 80, True, 0.1, None),
    (8, 1024, 132, True, 0.1, None), (8, 1024, 80, True, 0.1, None), (8, 1024, 512, False, 0.0, None),
    (2, 1000, 1000, True, 0.1, (256, 1)), (2, 1000, 1000, True, 0.1, (256, 3)), (3, 70, 513, True, 0.0, (256, 2)),
    (1, 200, 2048, False, 0.1, (256, 1)), (2, 1, 700, True, 0.1, (256, 1)), (2, 256, 600, True, 0.0, (128, 2)),
    (3, 33, 300, True, 0.1, (64, 1)), (2, 333, 77, True, 0.1, (64, 6)), (2, 130, 5, True, 0.1, (64, 3))])
def test_long_key_backward_equals_two_kernel_backward(mods, B, Lq, Lk, masked, p, force):
    """butd_attention_bwd_long_keys (one pass: a workgroup owns a chunk of 256 / 64 keys and walks its share of the
    queries, every score tile computed once; dQ through per-chunk slabs and, when the queries are split over several
    workgroups, dK / dV through per-split slabs, folded in order by one element-wise launch; delta formed while
    staging) against butd_attention_bwd on the same saved forward: same dropout masks (same hash), equal to fp32
    summation order (1e-5 of the scale), padding columns of the packed gradients untouched, and bit-reproducible.
    ``force``: (keys per workgroup, query splits) through the tuning hook, for shapes / plans the rule does not pick."""
    _, fa, _, _ = mods
    from butd_detr_amd import _hiplib
    lib = _hiplib.load()
    torch.manual_seed(Lq * 3 + Lk)
    H, D = 8, 36
    E = H * D
    dev = "cuda"
    q, do = torch.randn(B, Lq, E, device=dev) / 6, torch.randn(B, Lq, E, device=dev)
    k, v = torch.randn(B, Lk, E, device=dev), torch.randn(B, Lk, E, device=dev)
    mask = None
    if masked:
        mask = torch.zeros(B, Lk, dtype=torch.uint8, device=dev)
        for b in range(B):
            mask[b, Lk - 1 - min(40 * b, Lk // 2):] = 1
            if Lk >= 512:
                mask[b, 100:100 + 150 * (b % 2)] = 1      # a whole 16-key sub-tile and more in the middle
    out, lse = torch.empty_like(q), torch.empty(B, H, Lq, device=dev)
    ctr = fa.rng_counter(torch.device("cuda", 0))
    ctr.fill_(78)
    st = torch.cuda.current_stream().cuda_stream
    mp = mask.data_ptr() if mask is not None else None
    assert lib.butd_attention_fwd(B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), mp, out.data_ptr(),
                                  lse.data_ptr(), p, 5, ctr.data_ptr(), st) == 0
    ldq, ldkv = E + 8, 2 * E + 4
    assert lib.butd_attention_bwd_long_keys_scratch(1, 1, 64, 64, D, ldq) == -1        # would leave the part idle
    assert lib.butd_attention_bwd_long_keys_scratch(B, H, Lq, Lk, 32, ldq) == -1       # other head dimensions: not served
    assert lib.butd_attention_bwd_long_keys_scratch(B, H, Lq, Lk, D, E + 2) == -1      # dq rows not 16-byte aligned
    if force is not None:
        lib.butd_attention_bwd_long_keys_set_chunk(*force)
    try:
        need = int(lib.butd_attention_bwd_long_keys_scratch(B, H, Lq, Lk, D, ldq))
        assert need >= 0
        if force is not None:
            chunks, tiles = (Lk + force[0] - 1) // force[0], (Lq + 63) // 64
            per = (tiles + min(force[1], tiles) - 1) // min(force[1], tiles)
            splits = (tiles + per - 1) // per
            assert need == (chunks * B * Lq * E if chunks > 1 else 0) + (2 * splits * B * Lk * E if splits > 1 else 0)
        res = {}
        for name in ("two", "one", "one again"):
            dqb = torch.full((B, Lq, ldq), 7.0, device=dev)       # dq rows wider than E
            G = torch.full((B, Lk, ldkv), 7.0, device=dev)        # dk | dv side by side, rows wider than 2E
            if name == "two":
                delta = torch.empty(B, H, Lq, device=dev)
                assert lib.butd_attention_bwd(B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), mp, out.data_ptr(),
                                              do.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqb.data_ptr(), G.data_ptr(),
                                              G.data_ptr() + 4 * E, ldq, ldkv, 0.5, p, 5, ctr.data_ptr(), st) == 0
            else:
                ws = torch.full((max(need, 1),), float("nan"), device=dev)
                args = (B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), mp, out.data_ptr(), do.data_ptr(),
                        lse.data_ptr(), dqb.data_ptr(), G.data_ptr(), G.data_ptr() + 4 * E, ldq, ldkv, 0.5, p, 5,
                        ctr.data_ptr(), ws.data_ptr())
                assert lib.butd_attention_bwd_long_keys(*args, need, st) == 0
                if need > 0:
                    assert lib.butd_attention_bwd_long_keys(*args, need - 1, st) != 0      # workspace too small
            torch.cuda.synchronize()
            res[name] = (dqb[:, :, :E].clone(), G[:, :, :E].clone(), G[:, :, E:2 * E].clone(), dqb[:, :, E:].clone(),
                         G[:, :, 2 * E:].clone())
    finally:
        lib.butd_attention_bwd_long_keys_set_chunk(0, 0)
    for a, bq, name in zip(res["one"], res["two"], ("dq", "dk", "dv", "dq padding", "dkv padding")):
        assert torch.isfinite(a).all(), name
        if "padding" in name:
            assert torch.equal(a, bq) and float(a.min()) == 7.0, name
        else:
            _close(a, bq, 1e-5)
    for a, bq in zip(res["one"], res["one again"]):
        assert torch.equal(a, bq)


def test_long_key_backward_serves_the_blocks(mods):
    """The structured block on 1024 keys takes the one-pass backward by default (BUTD_AB=attn_long_keys=0 is the A/B
    switch) and its gradients equal the two-kernel walk's to fp32 summation order."""
    ab, fa, _, _ = mods
    from butd_detr_amd.encoder_decoder_layers import PosTransformerEncoderLayerNoFFN
    dev = torch.device("cuda", 0)
    torch.manual_seed(5)
    layer = PosTransformerEncoderLayerNoFFN(288, 8, dropout=0.1).to(dev).train()
    B, L = 8, 1024
    x, pos, probe = (torch.randn(B, L, 288, device=dev) for _ in range(3))
    ctr = fa.rng_counter(dev)

    def run(flag):
        prev = fa.set_long_keys(flag)
        ab.set_backend("hip")
        try:
            ctr.fill_(5)
            fa._site[0] = 0
            for prm in layer.parameters():
                prm.grad = None
            xi = x.clone().requires_grad_(True)
            out = layer(xi, pos)
            (out * probe).sum().backward()
            torch.cuda.synchronize()
            return [xi.grad.clone()] + [prm.grad.clone() for prm in layer.parameters() if prm.grad is not None]
        finally:
            ab.set_backend("torch")
            fa.set_long_keys(prev)

    assert fa._long_keys[0]
    one = run(True)
    two = run(False)
    assert len(one) == len(two) and len(one) > 4
    for a, b in zip(one, two):
        _close(a, b, 2e-5)


@pytest.mark.parametrize("rows,E", [(2048, 288), (8192, 288), (640, 288), (64, 32), (2048, 64)])
def test_layernorm_backward_partials_equal_the_atomic_kernel(mods, rows, E):
    """butd_add_dropout_layernorm_bwd_partial + the fold (ones . partials as one problem of the grouped product, what every
    block's backward appends to its next launch) against the kernel that accumulates dgamma / dbeta with atomics, and
    against float64; dx / d_residual identical."""
    _, fa, _, _ = mods
    from butd_detr_amd import _hiplib
    lib = _hiplib.load()
    torch.manual_seed(rows + E)
    dev = torch.device("cuda", 0)
    x, res, dy = (torch.randn(rows, E, device=dev) for _ in range(3))
    gamma = torch.randn(E, device=dev)
    s = x + res
    mean, var = s.mean(1), s.var(1, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    st = torch.cuda.current_stream().cuda_stream
    ctr = fa.rng_counter(dev).data_ptr()
    dx_a, dr_a, dx_p, dr_p = (torch.empty(rows, E, device=dev) for _ in range(4))
    gb_a = torch.zeros(2 * E, device=dev)
    assert lib.butd_add_dropout_layernorm_bwd(rows, E, dy.data_ptr(), x.data_ptr(), res.data_ptr(), gamma.data_ptr(),
                                              mean.data_ptr(), rstd.data_ptr(), dx_a.data_ptr(), dr_a.data_ptr(),
                                              gb_a.data_ptr(), gb_a.data_ptr() + 4 * E, 0.0, 3, ctr, st) == 0
    nb = lib.butd_layernorm_bwd_blocks(rows)
    assert nb >= 1 and nb % 4 == 0
    part = torch.full((nb, 2 * E), float("nan"), device=dev)        # fully overwritten
    assert lib.butd_add_dropout_layernorm_bwd_partial(rows, E, dy.data_ptr(), x.data_ptr(), res.data_ptr(),
                                                      gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                      dx_p.data_ptr(), dr_p.data_ptr(), part.data_ptr(), 0.0, 3, ctr, st) == 0
    gb_p = torch.full((2 * E,), float("nan"), device=dev)
    fa._gemm([fa._problem(fa._ones_row(dev, nb), part, gb_p, 1, 2 * E, nb, (nb, 1), (1, 2 * E), 2 * E)], x)
    torch.cuda.synchronize()
    assert torch.equal(dx_a, dx_p) and torch.equal(dr_a, dr_p)
    xhat = ((s - mean[:, None]) * rstd[:, None]).double()
    want = torch.cat([(dy.double() * xhat).sum(0), dy.double().sum(0)])
    scale = float(want.abs().max())
    assert float((gb_p.double() - want).abs().max()) <= 2e-6 * scale * max(1.0, (rows / 2048) ** 0.5)
    assert float((gb_p - gb_a).abs().max()) <= 1e-5 * scale
    assert float((part.sum(0).double() - want).abs().max()) <= 2e-6 * scale * max(1.0, (rows / 2048) ** 0.5)


def test_blocks_with_and_without_the_layernorm_fold_agree_and_the_fold_is_reproducible(mods):
    ab, fa, _, make_ffn = mods
    torch.manual_seed(3)
    E = 288
    ffn = make_ffn(E, 256, 0.1).cuda().eval()
    norm = torch.nn.LayerNorm(E).cuda()
    x = torch.randn(8, 256, E, device="cuda", requires_grad=True)
    probe = torch.randn(8, 256, E, device="cuda")

    def run():
        for t in [x] + list(norm.parameters()) + list(ffn.parameters()):
            t.grad = None
        (fa.ffn_block(ffn, norm, x) * probe).sum().backward()
        return [t.grad.clone() for t in [x] + list(norm.parameters()) + list(ffn.parameters())]

    prev = fa.set_ln_fold(True)
    try:
        g1, g2 = run(), run()
        fa.set_ln_fold(False)
        g0 = run()
    finally:
        fa.set_ln_fold(prev)
    assert torch.equal(g1[1], g2[1]) and torch.equal(g1[2], g2[2])      # dgamma, dbeta: no atomics, bit-reproducible
    for a, b in zip(g1, g0):
        _close(a, b, 1e-5)
