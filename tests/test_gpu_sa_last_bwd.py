"""-m gpu: the last shared-MLP layer + max-pool backward by linearity (include/butd_sa.h butd_sa_last_bwd,
csrc/sa_last_bwd.hip) vs (i) the dense path it replaces (butd_sa_dz_last + the two products + butd_sa_mask_stats) on the
same inputs and (ii) a float64 run of the stock module (pointnet2_modules.py:243-257, pytorch_utils.py:11-36).
fp32 throughout: north_star's tolerance is 1e-3; the two fp32 paths agree to ~1e-5 of each tensor's scale, and the
linear path is the closer one to float64 (it never rounds a dense dZ3)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _err(a, b):
    a, b = a.detach().double().cpu().numpy(), b.detach().double().cpu().numpy()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _stats(a, b):
    """(entries beyond 1e-3 of the scale, mean error / scale, size, worst): an arg-max or ReLU decision that falls the
    other way in fp32 than in float64 reroutes a few entries (as in test_gpu_fused_sa.py: 0.1 % allowed)."""
    a, b = a.detach().double().cpu().numpy(), b.detach().double().cpu().numpy()
    e = np.abs(a - b) / max(np.abs(b).max(), 1e-30)
    return int((e > 1e-3).sum()), float(e.mean()), e.size, float(e.max())


def _module(cfg, seed):
    from butd_detr_amd.pointnet2_modules import PointnetSAModuleVotes
    torch.manual_seed(seed)
    m = PointnetSAModuleVotes(npoint=cfg["npoint"], radius=cfg["radius"], nsample=cfg["nsample"],
                              mlp=list(cfg["mlp"]), use_xyz=True, normalize_xyz=True).cuda().train()
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.uniform_(-0.2, 0.2)
        m.mlp_module.layer2.bn.bn.weight[::7] *= -1.0      # negative scales on the LAST layer: the min-pool branch
        m.mlp_module.layer1.bn.bn.weight[:3] *= -1.0
    return m


def _run(m, xyz, feats, probe, linear, input_grad=True):
    from butd_detr_amd import attention_blocks, fused_sa
    prev = fused_sa.set_last_layer_linear(linear)
    attention_blocks.set_backend("hip")
    try:
        for p in m.parameters():
            p.grad = None
        f = feats.clone().requires_grad_(input_grad)
        y = m(xyz, f)[1]
        assert m.last_features_pm is not None, "fused path not taken"
        (y * probe).sum().backward()
        return (y.detach().clone(), f.grad.clone() if input_grad else None,
                {n: p.grad.clone() for n, p in m.named_parameters()})
    finally:
        attention_blocks.set_backend("torch")
        fused_sa.set_last_layer_linear(prev)


CFGS = [
    dict(B=2, N=4096, C=3, npoint=512, radius=0.4, nsample=64, mlp=[3, 64, 64, 128]),        # SA1-like
    dict(B=2, N=2048, C=128, npoint=1024, radius=0.6, nsample=32, mlp=[128, 128, 128, 256]),  # SA2
    dict(B=3, N=1024, C=256, npoint=512, radius=0.9, nsample=16, mlp=[256, 128, 128, 256]),   # SA3
    dict(B=1, N=300, C=256, npoint=3, radius=0.9, nsample=16, mlp=[256, 128, 128, 256]),      # 48 rows: a ragged block
    dict(B=1, N=700, C=3, npoint=5, radius=0.5, nsample=64, mlp=[3, 64, 64, 128]),            # 5 groups
]


@pytest.mark.parametrize("cfg", CFGS)
def test_linear_last_layer_equals_dense_path(cfg):
    m = _module(cfg, 3)
    torch.manual_seed(11)
    xyz = torch.rand(cfg["B"], cfg["N"], 3, device="cuda") * 2 - 1
    feats = torch.randn(cfg["B"], cfg["C"], cfg["N"], device="cuda")
    probe = torch.randn(cfg["B"], cfg["mlp"][-1], cfg["npoint"], device="cuda")
    y_d, gf_d, gp_d = _run(m, xyz, feats, probe, linear=False)
    y_l, gf_l, gp_l = _run(m, xyz, feats, probe, linear=True)
    # (the linear path's forward is butd_sa_last_fwd -- Z3 is never written --: same values to fp32 rounding; a pooled
    #  arg-max between two near-equal candidates may then fall the other way and reroute isolated gradient entries)
    assert _err(y_l, y_d) < 2e-6, _err(y_l, y_d)
    for n, a, b in [("d_feats", gf_l, gf_d)] + [(n, gp_l[n], gp_d[n]) for n in gp_d]:
        bad, mean, size, worst = _stats(a, b)
        assert bad <= max(2, 1e-3 * size) and mean < 2e-5 and worst < 2e-2, (n, bad, size, mean, worst)


@pytest.mark.parametrize("cfg", CFGS[:3])
def test_forward_without_z3_equals_product_plus_colstats(cfg):
    """butd_sa_last_fwd vs the layer's product launch + butd_sa_colstats it replaces (same backward): pooled output,
    BatchNorm running statistics, gradients."""
    from butd_detr_amd import fused_sa
    m = _module(cfg, 9)
    torch.manual_seed(17)
    xyz = torch.rand(cfg["B"], cfg["N"], 3, device="cuda") * 2 - 1
    feats = torch.randn(cfg["B"], cfg["C"], cfg["N"], device="cuda")
    probe = torch.randn(cfg["B"], cfg["mlp"][-1], cfg["npoint"], device="cuda")
    outs = {}
    state = {k: v.clone() for k, v in m.state_dict().items()}
    for fwd in (False, True):
        m.load_state_dict(state)
        prev = fused_sa._LAST_FWD[0]
        fused_sa._LAST_FWD[0] = fwd
        try:
            outs[fwd] = _run(m, xyz, feats, probe, linear=True) + ({k: v.clone() for k, v in m.state_dict().items()},)
        finally:
            fused_sa._LAST_FWD[0] = prev
    assert _err(outs[True][0], outs[False][0]) < 2e-6
    for k, v in outs[False][3].items():
        if "running" in k:
            assert _err(outs[True][3][k], v) < 1e-6, k
        if "num_batches" in k:
            assert int(outs[True][3][k]) == int(v)
    for n, a, b in [("d_feats", outs[True][1], outs[False][1])] + [(n, outs[True][2][n], outs[False][2][n]) for n in outs[False][2]]:
        bad, mean, size, worst = _stats(a, b)
        assert bad <= max(2, 1e-3 * size) and mean < 2e-5 and worst < 2e-2, (n, bad, size, mean, worst)


def test_linear_last_layer_is_bit_reproducible():
    """No atomics on the path: two runs give identical dW3 / BatchNorm sums of layer 2 (the dense path's split-K
    accumulation does not)."""
    cfg = CFGS[1]
    m = _module(cfg, 5)
    xyz = torch.rand(cfg["B"], cfg["N"], 3, device="cuda") * 2 - 1
    feats = torch.randn(cfg["B"], cfg["C"], cfg["N"], device="cuda")
    probe = torch.randn(cfg["B"], cfg["mlp"][-1], cfg["npoint"], device="cuda")
    a = _run(m, xyz, feats, probe, linear=True)[2]
    b = _run(m, xyz, feats, probe, linear=True)[2]
    for n in a:
        if "layer2" in n:
            assert torch.equal(a[n], b[n]), n


@pytest.mark.parametrize("cfg", CFGS[:3])
def test_linear_last_layer_vs_float64_module(cfg):
    """Against the stock module in float64 on the CPU with the SAME neighbour lists: both fp32 paths inside 1e-3
    (north_star), the linear one no further from the truth than the dense one (+ rounding slack)."""
    from butd_detr_amd import attention_blocks, pointnet2_utils
    m = _module(cfg, 7)
    torch.manual_seed(13)
    xyz = torch.rand(cfg["B"], cfg["N"], 3, device="cuda") * 2 - 1
    feats = torch.randn(cfg["B"], cfg["C"], cfg["N"], device="cuda")
    probe = torch.randn(cfg["B"], cfg["mlp"][-1], cfg["npoint"], device="cuda")
    y_d, gf_d, gp_d = _run(m, xyz, feats, probe, linear=False)
    y_l, gf_l, gp_l = _run(m, xyz, feats, probe, linear=True)
    # float64 truth: the module's own torch path with index ops taken from the fp32 GPU kernels (bit-exact vs the oracle)
    inds = pointnet2_utils.furthest_point_sample(xyz, cfg["npoint"])
    new_xyz = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    idx = pointnet2_utils.ball_query(cfg["radius"], cfg["nsample"], xyz, new_xyz).long().cpu()
    m.last_features_pm = None        # (a tensor with a grad_fn: not deep-copyable)
    m64 = copy.deepcopy(m).double().cpu()
    x64, f64 = xyz.double().cpu(), feats.double().cpu().requires_grad_(True)
    B, S, K = idx.shape
    bi = torch.arange(B)[:, None, None]
    grouped_xyz = (x64[bi, idx] - new_xyz.double().cpu()[:, :, None, :]) / cfg["radius"]       # (B, S, K, 3)
    grouped_f = f64.transpose(1, 2)[bi, idx]                                                     # (B, S, K, C)
    g = torch.cat([grouped_xyz, grouped_f], -1).permute(0, 3, 1, 2)                              # (B, 3+C, S, K)
    y64 = m64.mlp_module(g).max(-1)[0]
    (y64 * probe.double().cpu()).sum().backward()
    assert _err(y_l, y64) < 1e-4
    pairs = [("d_feats", gf_d, gf_l, f64.grad)]
    pairs += [(n, gp_d[n], gp_l[n], p.grad) for n, p in m64.named_parameters() if "mlp_module" in n and p.grad is not None]
    assert len(pairs) >= 8
    for n, dense, lin, truth in pairs:
        (_, m_d, _, _), (bad, m_l, size, worst) = _stats(dense, truth), _stats(lin, truth)
        assert bad <= max(2, 1e-3 * size) and worst < 2e-2 and m_l <= 1e-4, (n, bad, size, worst, m_l)
        assert m_l <= 1.5 * m_d + 1e-6, (n, m_l, m_d)        # typical error: no worse than the dense path


@pytest.mark.parametrize("cfg", [CFGS[0], CFGS[4], dict(B=8, N=20000, C=3, npoint=2048, radius=0.2, nsample=64, mlp=[3, 64, 64, 128])])
def test_first_layer_without_input_gradient(cfg):
    """SA1's features carry no gradient: butd_sa_first_bwd (sums + dW1 from one pass, no dZ1) vs butd_sa_mask_stats +
    butd_sa_dz_mid + the thin weight-gradient product, and vs the gradients of the run that also asks for d_feats."""
    from butd_detr_amd import fused_sa
    m = _module(cfg, 21)
    torch.manual_seed(23)
    xyz = torch.rand(cfg["B"], cfg["N"], 3, device="cuda") * 2 - 1
    feats = torch.randn(cfg["B"], cfg["C"], cfg["N"], device="cuda")
    probe = torch.randn(cfg["B"], cfg["mlp"][-1], cfg["npoint"], device="cuda")
    outs = {}
    # False: mask_stats + dz_mid + thin product;  True: butd_sa_first_bwd;  "mid": butd_sa_mid_first_bwd (layers 2 AND 1 in
    # one pass over (g2, Z2, Z1, X): also no dZ2, no dH1)
    # "noz1": also the forward never writes Z1 (butd_sa_first_two_fwd: layer 1's BatchNorm sums from the moments of X, z1
    # formed on the fly forward and backward)
    for key, first, mid, noz1 in ((False, False, False, False), (True, True, False, False), ("mid", True, True, False),
                                  ("noz1", True, True, True)):
        prev = fused_sa._FIRST_LIN[0], fused_sa._MID_FIRST[0], fused_sa._NO_Z1[0]
        fused_sa._FIRST_LIN[0], fused_sa._MID_FIRST[0], fused_sa._NO_Z1[0] = first, mid, noz1
        try:
            state = {k: v.clone() for k, v in m.state_dict().items()}
            outs[key] = _run(m, xyz, feats, probe, linear=True, input_grad=False) + ({k: v.clone() for k, v in m.state_dict().items()},)
            m.load_state_dict(state)
        finally:
            fused_sa._FIRST_LIN[0], fused_sa._MID_FIRST[0], fused_sa._NO_Z1[0] = prev
    prev = fused_sa._NO_Z1[0]
    fused_sa._NO_Z1[0] = False
    try:
        ref = _run(m, xyz, feats, probe, linear=True, input_grad=True)
    finally:
        fused_sa._NO_Z1[0] = prev
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs["mid"][0], outs[False][0])
    assert _err(outs["noz1"][0], outs[False][0]) < 2e-6
    for k, v in outs[False][3].items():               # BatchNorm running statistics after the step
        if "running" in k:
            assert _err(outs["noz1"][3][k], v) < 1e-5, k
    for n in outs[False][2]:
        for k in (True, "mid"):
            assert _err(outs[k][2][n], outs[False][2][n]) < 2e-5, (k, n, _err(outs[k][2][n], outs[False][2][n]))
            assert _err(outs[k][2][n], ref[2][n]) < 2e-5, (k, n)
        bad, mean, size, worst = _stats(outs["noz1"][2][n], outs[False][2][n])       # (another forward rounding: reroutes allowed)
        assert bad <= max(2, 1e-3 * size) and mean < 2e-5 + 2 * worst / size and worst < 2e-2, (n, bad, size, mean, worst)


@pytest.mark.parametrize("cfg", CFGS[1:4])
def test_feature_gradient_as_gather_equals_scatter(cfg):
    """butd_sa_inverse_index + butd_sa_gather_rows vs butd_sa_scatter_rows (float atomics), and the inverse lists
    themselves: every grouped row appears exactly once, in the slice of the point it copies."""
    from butd_detr_amd import fused_sa, pointnet2_utils
    m = _module(cfg, 31)
    torch.manual_seed(37)
    xyz = torch.rand(cfg["B"], cfg["N"], 3, device="cuda") * 2 - 1
    feats = torch.randn(cfg["B"], cfg["C"], cfg["N"], device="cuda")
    probe = torch.randn(cfg["B"], cfg["mlp"][-1], cfg["npoint"], device="cuda")
    outs = {}
    for gather in (False, True):
        prev = fused_sa._GATHER[0]
        fused_sa._GATHER[0] = gather
        try:
            outs[gather] = _run(m, xyz, feats, probe, linear=True)
        finally:
            fused_sa._GATHER[0] = prev
    assert _err(outs[True][1], outs[False][1]) < 1e-5
    inds = pointnet2_utils.furthest_point_sample(xyz, cfg["npoint"])
    new_xyz = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    idx = pointnet2_utils.ball_query(cfg["radius"], cfg["nsample"], xyz, new_xyz)
    start, lst = fused_sa.inverse_index(idx, cfg["N"])
    start, lst, flat = start.cpu().numpy(), lst.cpu().numpy(), idx.reshape(cfg["B"], -1).cpu().numpy()
    P = flat.size
    assert start[0] == 0 and start[-1] == P and (np.diff(start) >= 0).all()
    assert np.array_equal(np.sort(lst), np.arange(P))
    owner = np.repeat(np.arange(cfg["B"] * cfg["N"]), np.diff(start))        # the point each list entry is filed under
    per = flat.shape[1]
    assert np.array_equal(owner, (lst // per) * cfg["N"] + flat.reshape(-1)[lst])
    assert all((np.diff(lst[a:b]) > 0).all() for a, b in zip(start[:-1], start[1:]))      # ascending slices: unique result


@pytest.mark.parametrize("cfg", CFGS[1:4] + [dict(B=8, N=2048, C=128, npoint=1024, radius=0.4, nsample=32, mlp=[128, 128, 128, 256])])
def test_wide_middle_layer_in_one_pass(cfg):
    """SA2-4 (128-wide layers, input gradient wanted): butd_sa_mid_wide_bwd (dW2, layer 1's gated gradient and its sums
    from one pass over (g2, Z2, Z1)) vs butd_sa_dz_mid + the product pair with the statistics epilogue: same forward, so
    every gradient agrees to fp32 rounding; the one-pass kernel has no atomics: bit-reproducible dW2."""
    from butd_detr_amd import fused_sa
    m = _module(cfg, 41)
    torch.manual_seed(43)
    xyz = torch.rand(cfg["B"], cfg["N"], 3, device="cuda") * 2 - 1
    feats = torch.randn(cfg["B"], cfg["C"], cfg["N"], device="cuda")
    probe = torch.randn(cfg["B"], cfg["mlp"][-1], cfg["npoint"], device="cuda")
    outs = {}
    for key, wide in (("pair", False), ("wide", True), ("wide again", True)):
        prev = fused_sa._MID_WIDE[0]
        fused_sa._MID_WIDE[0] = wide
        try:
            state = {k: v.clone() for k, v in m.state_dict().items()}
            outs[key] = _run(m, xyz, feats, probe, linear=True)
            m.load_state_dict(state)
        finally:
            fused_sa._MID_WIDE[0] = prev
    assert torch.equal(outs["wide"][0], outs["pair"][0])
    assert _err(outs["wide"][1], outs["pair"][1]) < 2e-5, _err(outs["wide"][1], outs["pair"][1])
    for n in outs["pair"][2]:
        assert _err(outs["wide"][2][n], outs["pair"][2][n]) < 2e-5, (n, _err(outs["wide"][2][n], outs["pair"][2][n]))
    assert torch.equal(outs["wide"][2]["mlp_module.layer1.conv.weight"], outs["wide again"][2]["mlp_module.layer1.conv.weight"])


def _with_first_linear(flag, fn):
    from butd_detr_amd import fused_sa
    prev = fused_sa._FIRST_LINEAR[0]
    fused_sa._FIRST_LINEAR[0] = flag
    try:
        return fn()
    finally:
        fused_sa._FIRST_LINEAR[0] = prev


@pytest.mark.parametrize("cfg", CFGS[1:4] + [dict(B=8, N=2048, C=128, npoint=1024, radius=0.4, nsample=32, mlp=[128, 128, 128, 256]),
                                             dict(B=2, N=900, C=64, npoint=300, radius=0.7, nsample=16, mlp=[64, 64, 128, 128])])
@pytest.mark.parametrize("input_grad", [True, False])
def test_first_layer_by_linearity_equals_grouped_input_path(cfg, input_grad):
    """Round 6 (csrc/sa_first_linear.hip): levels with input features run their first layer over the N points of the level
    (Y = feats Wf^T, Z1 = gather + the three coordinate terms) and its backward along the inverted neighbour lists (T, dWx;
    d_feats = T Wf, dWf = T^T feats) -- vs the path that forms the grouped input X (butd_sa_group + the P-row products +
    butd_sa_dz_mid + butd_sa_gather_rows).  Same arithmetic up to the order in which a Z1 element's 3 + C terms are summed."""
    m = _module(cfg, 51)
    torch.manual_seed(53)
    xyz = torch.rand(cfg["B"], cfg["N"], 3, device="cuda") * 2 - 1
    feats = torch.randn(cfg["B"], cfg["C"], cfg["N"], device="cuda")
    probe = torch.randn(cfg["B"], cfg["mlp"][-1], cfg["npoint"], device="cuda")
    outs = {}
    for key, flag in (("grouped", False), ("linear", True), ("linear again", True)):
        state = {k: v.clone() for k, v in m.state_dict().items()}
        outs[key] = _with_first_linear(flag, lambda: _run(m, xyz, feats, probe, linear=True, input_grad=input_grad))
        if key == "grouped":
            buffers = {k: v.clone() for k, v in m.state_dict().items() if "running" in k}
        else:
            for k, v in buffers.items():          # the running statistics of the three layers after one training step
                assert _err(m.state_dict()[k], v) < 1e-5, k
        m.load_state_dict(state)
    # a ReLU gate or an arg-max that falls the other way after the reassociation reroutes single entries
    bad, mean, size, worst = _stats(outs["linear"][0], outs["grouped"][0])
    assert mean < 1e-6 and bad <= max(1, 1e-4 * size), (bad, mean, size, worst)
    if input_grad:
        bad, mean, size, worst = _stats(outs["linear"][1], outs["grouped"][1])
        assert mean < 2e-6 and bad <= max(2, 1e-3 * size), (bad, mean, size, worst)
    for n in outs["grouped"][2]:
        bad, mean, size, worst = _stats(outs["linear"][2][n], outs["grouped"][2][n])
        assert mean < 1e-4 and bad <= max(2, 2e-3 * size) and worst < 1e-2, (n, bad, mean, size, worst)   # (vs float64: the test below)
    # T and dWx are gathered / summed in a fixed order; the products behind them use the step's float atomics
    assert torch.equal(outs["linear"][0], outs["linear again"][0])


@pytest.mark.parametrize("cfg", CFGS[1:3])
def test_first_layer_by_linearity_vs_float64_module(cfg):
    """The same level against the stock module in float64 (same neighbour lists): the linear first layer is no further
    from the truth than the grouped-input path, both inside north_star's 1e-3."""
    from butd_detr_amd import pointnet2_utils
    m = _module(cfg, 57)
    torch.manual_seed(59)
    xyz = torch.rand(cfg["B"], cfg["N"], 3, device="cuda") * 2 - 1
    feats = torch.randn(cfg["B"], cfg["C"], cfg["N"], device="cuda")
    probe = torch.randn(cfg["B"], cfg["mlp"][-1], cfg["npoint"], device="cuda")
    state = {k: v.clone() for k, v in m.state_dict().items()}
    y_g, gf_g, gp_g = _with_first_linear(False, lambda: _run(m, xyz, feats, probe, linear=True))
    m.load_state_dict(state)
    y_l, gf_l, gp_l = _with_first_linear(True, lambda: _run(m, xyz, feats, probe, linear=True))
    m.load_state_dict(state)
    inds = pointnet2_utils.furthest_point_sample(xyz, cfg["npoint"])
    new_xyz = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    idx = pointnet2_utils.ball_query(cfg["radius"], cfg["nsample"], xyz, new_xyz).long().cpu()
    m.last_features_pm = None
    m64 = copy.deepcopy(m).double().cpu()
    x64, f64 = xyz.double().cpu(), feats.double().cpu().requires_grad_(True)
    bi = torch.arange(idx.shape[0])[:, None, None]
    grouped_xyz = (x64[bi, idx] - new_xyz.double().cpu()[:, :, None, :]) / cfg["radius"]
    g = torch.cat([grouped_xyz, f64.transpose(1, 2)[bi, idx]], -1).permute(0, 3, 1, 2)
    y64 = m64.mlp_module(g).max(-1)[0]
    (y64 * probe.double().cpu()).sum().backward()
    assert _err(y_l, y64) < 1e-4
    pairs = [("d_feats", gf_g, gf_l, f64.grad)]
    pairs += [(n, gp_g[n], gp_l[n], p.grad) for n, p in m64.named_parameters() if "mlp_module" in n and p.grad is not None]
    for n, grouped, lin, truth in pairs:
        (_, m_g, _, _), (bad, m_l, size, worst) = _stats(grouped, truth), _stats(lin, truth)
        assert bad <= max(2, 1e-3 * size) and worst < 2e-2 and m_l <= 1e-4, (n, bad, size, worst, m_l)
        assert m_l <= 1.5 * m_g + 1e-6, (n, m_l, m_g)
