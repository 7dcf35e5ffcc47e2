"""-m gpu: the Conv1d+BatchNorm1d+ReLU(+Dropout) chains on the grouped-GEMM pipeline of include/butd_mlp.h
(fused_mlp.mlp_chains) vs the stock-torch modules, forward and backward, train and eval
(models/modules.py:19-180 semantics).  Dropout is checked for determinism and fwd/bwd mask consistency."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(a, b, tol=1e-3, frac=0.0):
    """|a-b| <= tol * max|b| everywhere, except a fraction ``frac`` of outliers (a pre-activation within
    rounding distance of 0 switches its ReLU on one side only)."""
    a, b = a.detach().float().cpu().numpy(), b.detach().float().cpu().numpy()
    scale = max(np.abs(b).max(), 1e-6)
    bad = np.abs(a - b) / scale > tol
    assert bad.mean() <= frac, f"{bad.sum()}/{bad.size} beyond {tol}; max {np.abs(a - b).max() / scale:.3e}"


def _randomize_bn(mod):
    with torch.no_grad():
        for m in mod.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.2, 0.2)
                m.running_mean.uniform_(-0.1, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight[:3] *= -1.0


def _set_dropout(mod, p):
    for m in mod.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = p


@pytest.fixture
def backend():
    from butd_detr_amd import attention_blocks as ab
    yield ab
    ab.set_backend("torch")


def _compare(ab, ref, fused, run, train, zero_grads=()):
    """``zero_grads``: parameters whose gradient is exactly 0 in exact arithmetic when training (a conv
    bias in front of a batch-statistics BatchNorm): both sides return rounding noise, only its size is
    checked."""
    ref.train(train)
    fused.train(train)
    ab.set_backend("torch")
    out_r, gin_r = run(ref)
    ab.set_backend("hip")
    out_f, gin_f = run(fused)
    ab.set_backend("torch")
    for a, b in zip(out_f, out_r):
        assert a.shape == b.shape
        _close(a, b)
    for a, b in zip(gin_f, gin_r):
        _close(a, b, 2e-3, frac=1e-3)
    gmax = max(float(p.grad.abs().max()) for p in ref.parameters() if p.grad is not None)
    for (n, pf), (_, pr) in zip(fused.named_parameters(), ref.named_parameters()):
        assert (pf.grad is None) == (pr.grad is None), n
        if pr.grad is None:
            continue
        if train and n in zero_grads:
            assert float(pf.grad.abs().max()) <= 1e-3 * gmax, n
        else:
            _close(pf.grad, pr.grad, 2e-3, frac=1e-3)
    for (n, bf), (_, br) in zip(fused.named_buffers(), ref.named_buffers()):
        if bf.dtype.is_floating_point:
            _close(bf, br, 1e-4)
        else:
            assert torch.equal(bf, br), n


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("B,Q", [(8, 256), (2, 100)])
def test_predict_head_matches_torch(backend, train, B, Q):
    from butd_detr_amd.modules import ClsAgnosticPredictHead
    torch.manual_seed(Q + train)
    ref = ClsAgnosticPredictHead(256, 1, Q, 288, objectness=False, heading=False,
                                 compute_sem_scores=True).cuda()
    _randomize_bn(ref)
    _set_dropout(ref, 0.0)
    fused = copy.deepcopy(ref)
    feats = torch.randn(B, Q, 288, device="cuda")
    base = torch.randn(B, Q, 3, device="cuda")
    probes = [torch.randn(B, Q, 3, device="cuda"), torch.randn(B, Q, 3, device="cuda"),
              torch.randn(B, Q, 256, device="cuda")]

    def run(mod):
        x = feats.clone().requires_grad_(True)
        ep = {}
        use_pm = mod is fused
        center, size = mod(x.transpose(1, 2) if use_pm else x.transpose(1, 2).contiguous(), base, ep,
                           prefix="t_", **({"features_pm": x} if use_pm else {}))
        outs = [center, size, ep["t_sem_cls_scores"]]
        loss = sum((o * p).sum() for o, p in zip(outs, probes))
        loss.backward()
        return outs, [x.grad]

    _compare(backend, ref, fused, run, train)


@pytest.mark.parametrize("train", [True, False])
def test_predict_head_with_objectness(backend, train):
    from butd_detr_amd.modules import ClsAgnosticPredictHead
    torch.manual_seed(5)
    ref = ClsAgnosticPredictHead(18, 1, 64, 128, objectness=True, heading=False,
                                 compute_sem_scores=True).cuda()
    _randomize_bn(ref)
    _set_dropout(ref, 0.0)
    fused = copy.deepcopy(ref)
    feats = torch.randn(3, 128, 64, device="cuda")
    base = torch.randn(3, 64, 3, device="cuda")

    def run(mod):
        x = feats.clone().requires_grad_(True)
        ep = {}
        center, size = mod(x, base, ep, prefix="")
        outs = [center, size, ep["sem_cls_scores"], ep["objectness_scores"]]
        sum((o * o).sum() for o in outs).backward()
        return outs, [x.grad]

    _compare(backend, ref, fused, run, train)


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("cin,feat,B,N", [(6, 288, 8, 256), (3, 288, 2, 1024), (6, 128, 4, 132)])
def test_position_embedding_matches_torch(backend, train, cin, feat, B, N):
    from butd_detr_amd.encoder_decoder_layers import PositionEmbeddingLearned
    torch.manual_seed(cin + N)
    ref = PositionEmbeddingLearned(cin, feat).cuda()
    _randomize_bn(ref)
    fused = copy.deepcopy(ref)
    xyz = torch.randn(B, N, cin, device="cuda")

    def run(mod):
        x = xyz.clone().requires_grad_(True)
        out = mod(x)
        (out * out).sum().backward()
        return [out], [x.grad]

    _compare(backend, ref, fused, run, train, zero_grads=("position_embedding_head.0.bias",))


@pytest.mark.parametrize("train", [True, False])
def test_points_obj_cls_matches_torch(backend, train):
    from butd_detr_amd.modules import PointsObjClsModule
    torch.manual_seed(2)
    ref = PointsObjClsModule(288).cuda()
    _randomize_bn(ref)
    fused = copy.deepcopy(ref)
    feats = torch.randn(4, 288, 1024, device="cuda")

    def run(mod):
        x = feats.clone().requires_grad_(True)
        out = mod(x)
        (out * out).sum().backward()
        return [out], [x.grad]

    _compare(backend, ref, fused, run, train, zero_grads=("conv1.bias", "conv2.bias"))


def test_three_layer_mlp_standalone_and_dropout(backend):
    from butd_detr_amd import fused_attention as fa
    from butd_detr_amd.modules import ThreeLayerMLP
    torch.manual_seed(0)
    mlp = ThreeLayerMLP(288, 7).cuda().train()
    _randomize_bn(mlp)
    x = torch.randn(4, 288, 200, device="cuda", requires_grad=True)
    dev = x.device
    backend.set_backend("hip")
    state = {n: b.clone() for n, b in mlp.named_buffers()}

    def f(t):
        with torch.no_grad():            # running statistics do not drift between the probes
            for n, b in mlp.named_buffers():
                b.copy_(state[n])
        fa._site[0] = 300                # same sites -> same masks within a step
        return mlp(t)

    fa.new_step(dev)
    y1, y2 = f(x), f(x)
    assert y1.shape == (4, 7, 200) and torch.equal(y1, y2)
    fa.new_step(dev)
    assert not torch.equal(y1, f(x))     # a new step draws new masks
    y1 = f(x)

    # exact parity against a torch restatement that is handed the SAME masks: butd_mlp_mask_stats on
    # dH = 1 with an always-positive pre-activation returns keep/(1-p) of (site, element)
    from butd_detr_amd import _hiplib
    lib = _hiplib.load()
    P, H = 4 * 200, 288
    masks = []
    for site in (301, 302):              # site0 = _site + 1, one site per hidden layer (G = 1)
        m = torch.ones(P, H, device=dev)
        zeros, ones = torch.zeros(P, H, device=dev), torch.ones(H, device=dev)
        S = torch.zeros(2, H, dtype=torch.float64, device=dev)
        err = lib.butd_mlp_mask_stats(P, H, H, m.data_ptr(), zeros.data_ptr(), ones.data_ptr(), ones.data_ptr(),
                                      zeros.data_ptr(), ones.data_ptr(), 0.3, site, H,
                                      fa.rng_counter(dev).data_ptr(), S[0].data_ptr(), S[1].data_ptr(),
                                      torch.cuda.current_stream().cuda_stream)
        assert err == 0
        keep = (m > 0).float().mean().item()
        assert abs(keep - 0.7) < 0.01, keep
        assert torch.all((m == 0) | ((m - 1 / 0.7).abs() < 1e-6))
        masks.append(m.view(4, 200, H).transpose(1, 2))
    net = mlp.net

    def emulated(t):
        h = torch.relu(net[1](net[0](t))) * masks[0]
        h = torch.relu(net[5](net[4](h))) * masks[1]
        return net[8](h)

    probe = torch.randn_like(y1)
    x.grad = None
    mlp.zero_grad()
    (y1 * probe).sum().backward()
    got = [x.grad.clone()] + [p.grad.clone() for p in mlp.parameters()]
    x.grad = None
    mlp.zero_grad()
    backend.set_backend("torch")
    with torch.no_grad():
        for n, b in mlp.named_buffers():
            b.copy_(state[n])
    y_ref = emulated(x)
    (y_ref * probe).sum().backward()
    want = [x.grad.clone()] + [p.grad.clone() for p in mlp.parameters()]
    _close(y1, y_ref)
    for a, b in zip(got, want):
        _close(a, b, 2e-3, frac=1e-3)


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("mlp,n,m", [([512, 256, 256], 512, 256), ([512, 256, 288], 1024, 512)])
def test_fp_module_matches_torch(backend, train, mlp, n, m):
    """PointnetFPModule (pointnet2_modules.py:371-416): 3-NN interpolation (HIP index ops on both sides)
    + skip concat + SharedMLP -- the SharedMLP as a BatchNorm+ReLU-terminated fused chain vs stock torch."""
    from butd_detr_amd.pointnet2_modules import PointnetFPModule
    torch.manual_seed(n)
    B = 4
    ref = PointnetFPModule(mlp=list(mlp)).cuda()
    with torch.no_grad():
        for mod in ref.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.uniform_(-0.2, 0.2)
                mod.running_mean.uniform_(-0.1, 0.1)
                mod.running_var.uniform_(0.5, 1.5)
                mod.weight[:3] *= -1.0
    fused = copy.deepcopy(ref)
    unknown = torch.rand(B, n, 3, device="cuda")
    known = torch.rand(B, m, 3, device="cuda")
    c_skip, c_known = mlp[0] // 2, mlp[0] - mlp[0] // 2
    f_unknown = torch.randn(B, c_skip, n, device="cuda")
    f_known = torch.randn(B, c_known, m, device="cuda")
    probe = torch.randn(B, mlp[-1], n, device="cuda")

    def run(mod):
        a = f_unknown.clone().requires_grad_(True)
        k = f_known.clone().requires_grad_(True)
        out = mod(unknown, known, a, k)
        (out * probe).sum().backward()
        return [out], [a.grad, k.grad]

    _compare(backend, ref, fused, run, train)


@pytest.mark.parametrize("train,p_drop", [(True, 0.0), (True, 0.3), (False, 0.0)])
@pytest.mark.parametrize("B,Q", [(8, 256), (2, 100)])
def test_gate_and_statistics_in_the_product_epilogue_equal_the_separate_pass(backend, train, p_drop, B, Q):
    """butd_gemm_problem.c_bn_* (the product that creates a hidden gradient applies the BatchNorm + ReLU gate and leaves
    the two column sums of the BatchNorm backward) against butd_mlp_mask_stats as a pass of its own: same gradients for
    a whole predict head (three chains; the 3-wide box heads go through the element-wise staged kernel, the 256-wide
    class head through the float4 one; (2, 100): ragged row tiles; p 0.3: the reference's Dropout behind every hidden
    activation, its mask regenerated inside the epilogue)."""
    from butd_detr_amd import fused_mlp
    from butd_detr_amd.modules import ClsAgnosticPredictHead
    backend.set_backend("hip")               # (the fixture only restores the stock backend afterwards)
    torch.manual_seed(Q)
    head = ClsAgnosticPredictHead(256, 1, Q, 288, objectness=True, heading=False, compute_sem_scores=True).cuda()
    _randomize_bn(head)
    _set_dropout(head, p_drop)
    head.train(train)
    feats = torch.randn(B, Q, 288, device="cuda")
    base = torch.randn(B, Q, 3, device="cuda")
    probes = [torch.randn(B, Q, 3, device="cuda"), torch.randn(B, Q, 3, device="cuda"),
              torch.randn(B, Q, 256, device="cuda"), torch.randn(B, Q, device="cuda")]

    def run():
        from butd_detr_amd import fused_attention
        fused_attention._site[0] = 0          # the same dropout sites (= the same masks) in both runs
        for p_ in head.parameters():
            p_.grad = None
        x = feats.clone().requires_grad_(True)
        ep = {}
        center, size = head(x.transpose(1, 2), base, ep, prefix="t_", features_pm=x)
        outs = [center, size, ep["t_sem_cls_scores"], ep["t_objectness_scores"]]
        sum((o * p_).sum() for o, p_ in zip(outs, probes)).backward()
        return [x.grad.clone()] + [p_.grad.clone() for p_ in head.parameters() if p_.grad is not None]

    prev = fused_mlp.set_fuse_stats(True)
    try:
        g_fused = run()
        fused_mlp.set_fuse_stats(False)
        g_sep = run()
    finally:
        fused_mlp.set_fuse_stats(prev)
    assert len(g_fused) == len(g_sep) > 10
    for a, b in zip(g_fused, g_sep):
        _close(a, b, 2e-5)
    if train and p_drop > 0:                 # the fused chains ran: a second step draws other masks
        from butd_detr_amd import fused_attention
        fused_attention.new_step(feats.device)
        assert not torch.equal(run()[0], g_sep[0])
