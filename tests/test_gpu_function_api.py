"""-m gpu: the pointnet2_utils Function / module API of boundary (b)-2 beyond the raw ops:
QueryAndGroup with every option, GroupAll, autograd contracts (pointnet2_utils.py:51-291, 306-426)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(seed=0, B=2, N=600, M=40, C=5):
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, rng.integers(0, N, M)])
    feats = rng.standard_normal((B, C, N)).astype(np.float32)
    return xyz, new_xyz, feats


def _expected(oracle, xyz, new_xyz, feats, radius, ns, normalize):
    idx = oracle.ball_query(new_xyz, xyz, radius, ns).astype(np.int64)            # (B,M,S)
    B, M, S = idx.shape
    g_xyz = np.stack([xyz[b][idx[b]] for b in range(B)])                          # (B,M,S,3)
    g_xyz = (g_xyz - new_xyz[:, :, None, :]).transpose(0, 3, 1, 2)                # (B,3,M,S)
    if normalize:
        g_xyz = g_xyz / np.float32(radius)
    g_f = np.stack([feats[b][:, idx[b]] for b in range(B)])                       # (B,C,M,S)
    return idx, g_xyz.astype(np.float32), g_f


@pytest.mark.parametrize("use_xyz,normalize,ret_xyz", [(True, False, False), (True, True, True), (False, False, True)])
def test_query_and_group_options(oracle, use_xyz, normalize, ret_xyz):
    from butd_detr_amd import pointnet2_utils as pu
    xyz, new_xyz, feats = _case()
    radius, ns = 0.35, 16
    idx, g_xyz, g_f = _expected(oracle, xyz, new_xyz, feats, radius, ns, normalize)
    mod = pu.QueryAndGroup(radius, ns, use_xyz=use_xyz, ret_grouped_xyz=ret_xyz, normalize_xyz=normalize)
    f = torch.from_numpy(feats).cuda().requires_grad_(True)
    out = mod(torch.from_numpy(xyz).cuda(), torch.from_numpy(new_xyz).cuda(), f)
    new_features, grouped_xyz = (out if ret_xyz else (out, None))
    want = np.concatenate([g_xyz, g_f], 1) if use_xyz else g_f                   # xyz channels FIRST (:361-364)
    np.testing.assert_allclose(new_features.detach().cpu().numpy(), want, rtol=0, atol=1e-6)
    if ret_xyz:
        np.testing.assert_allclose(grouped_xyz.cpu().numpy(), g_xyz, rtol=0, atol=1e-6)
    # gradient reaches the features only, as a scatter-add over the group indices (pointnet2_utils.py:236-257)
    probe = torch.randn_like(new_features)
    (new_features * probe).sum().backward()
    pf = probe[:, -feats.shape[1]:].cpu().numpy()
    want_grad = np.zeros_like(feats)
    for b in range(feats.shape[0]):
        np.add.at(want_grad[b], (slice(None), idx[b].reshape(-1)), pf[b].reshape(feats.shape[1], -1))
    np.testing.assert_allclose(f.grad.cpu().numpy(), want_grad, rtol=1e-4, atol=1e-4)
    # xyz only
    only = pu.QueryAndGroup(radius, ns, use_xyz=True, normalize_xyz=normalize)(
        torch.from_numpy(xyz).cuda(), torch.from_numpy(new_xyz).cuda(), None)
    np.testing.assert_allclose(only.cpu().numpy(), g_xyz, rtol=0, atol=1e-6)
    with pytest.raises(AssertionError):
        pu.QueryAndGroup(radius, ns, use_xyz=False)(torch.from_numpy(xyz).cuda(), torch.from_numpy(new_xyz).cuda(), None)


def test_query_and_group_sample_uniformly(oracle):
    """:336-345: the pad-with-first-hit tail is replaced by uniform draws from the ball's unique hits; the
    unique hits come first, `unique_cnt` counts them."""
    from butd_detr_amd import pointnet2_utils as pu
    xyz, new_xyz, feats = _case(seed=3, N=300, M=12)
    radius, ns = 0.3, 16
    ref_idx = oracle.ball_query(new_xyz, xyz, radius, ns)
    mod = pu.QueryAndGroup(radius, ns, use_xyz=False, sample_uniformly=True, ret_unique_cnt=True)
    marker = torch.arange(xyz.shape[1], dtype=torch.float32).repeat(xyz.shape[0], 1, 1).cuda()   # feature = index
    out, cnt = mod(torch.from_numpy(xyz).cuda(), torch.from_numpy(new_xyz).cuda(), marker)
    got = out[:, 0].cpu().numpy().astype(np.int64)                                              # (B,M,S) indices
    for b in range(got.shape[0]):
        for r in range(got.shape[1]):
            uniq = np.unique(ref_idx[b, r])
            assert int(cnt[b, r]) == len(uniq)
            np.testing.assert_array_equal(got[b, r, : len(uniq)], uniq)
            assert set(got[b, r].tolist()) <= set(uniq.tolist())
    with pytest.raises(AssertionError):
        pu.QueryAndGroup(radius, ns, ret_unique_cnt=True)


def test_group_all():
    from butd_detr_amd import pointnet2_utils as pu
    xyz, _, feats = _case(seed=5, N=50, C=4)
    x, f = torch.from_numpy(xyz).cuda(), torch.from_numpy(feats).cuda()
    out = pu.GroupAll(use_xyz=True)(x, None, f)
    assert out.shape == (2, 3 + 4, 1, 50)
    assert torch.equal(out[:, :3, 0], x.transpose(1, 2)) and torch.equal(out[:, 3:, 0], f)
    assert torch.equal(pu.GroupAll(use_xyz=False)(x, None, f), f.unsqueeze(2))
    assert torch.equal(pu.GroupAll()(x, None, None)[:, :, 0], x.transpose(1, 2))
    assert not isinstance(pu.GroupAll(ret_grouped_xyz=True)(x, None, f), tuple)    # forced off, pointnet2_utils.py:390


def test_index_outputs_are_not_differentiable():
    from butd_detr_amd import pointnet2_utils as pu
    xyz = torch.rand(2, 200, 3, device="cuda", requires_grad=True)
    inds = pu.furthest_point_sample(xyz, 16)
    assert inds.dtype == torch.int32 and not inds.requires_grad
    new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    assert new_xyz.requires_grad                                   # gather back-propagates to the features
    idx = pu.ball_query(0.3, 8, xyz, new_xyz)
    assert idx.dtype == torch.int32 and not idx.requires_grad
    dist, nn = pu.three_nn(xyz, new_xyz)
    assert not dist.requires_grad and not nn.requires_grad
