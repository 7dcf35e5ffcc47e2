"""-m gpu: the grid-pruned ball query (butd_ball_query_ws, include/butd_pointnet2.h) is bit-identical to the
CPU oracle (src/ball_query_gpu.cu:13-49 semantics: first `nsample` hits in ascending index order, padded
with the first hit, zero rows without a hit) and to the streaming kernel -- on uniform, surface-like and
clustered clouds, centres outside the cloud, NaN / inf coordinates, degenerate extents and radii."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from butd_detr_amd.synthetic_scenes import scene_batch  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def ws_bytes(b, n):
    return 24 * b * n + 264192 * b  # the bound include/butd_pointnet2.h documents


def pruned(new_xyz, xyz, radius, nsample):
    """butd_ball_query_ws with a workspace that forces the pruned path (any n <= 2^18)."""
    from butd_detr_amd import _hiplib
    lib = _hiplib.load()
    b, m, n = new_xyz.shape[0], new_xyz.shape[1], xyz.shape[1]
    idx = torch.full((b, m, nsample), -7, dtype=torch.int32, device="cuda")
    ws = torch.full((ws_bytes(b, n),), 255, dtype=torch.uint8, device="cuda")
    err = lib.butd_ball_query_ws(b, n, m, float(radius), nsample, new_xyz.data_ptr(), xyz.data_ptr(),
                                 idx.data_ptr(), ws.data_ptr(), ws.numel(),
                                 torch.cuda.current_stream().cuda_stream)
    assert err == 0
    torch.cuda.synchronize()
    # the workspace starts with the cell-sorted (x, y, z, index) records: the pruned path ran iff they exist
    assert sorted(ws[: 16 * n].view(torch.int32)[3::4].tolist()) == list(range(n))
    return idx


def streaming(new_xyz, xyz, radius, nsample):
    from butd_detr_amd import _hiplib
    lib = _hiplib.load()
    b, m, n = new_xyz.shape[0], new_xyz.shape[1], xyz.shape[1]
    idx = torch.full((b, m, nsample), -7, dtype=torch.int32, device="cuda")
    err = lib.butd_ball_query(b, n, m, float(radius), nsample, new_xyz.data_ptr(), xyz.data_ptr(),
                              idx.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert err == 0
    torch.cuda.synchronize()
    return idx


@pytest.mark.parametrize("n,m,ns,r", [(1, 1, 4, 0.5), (63, 5, 8, 0.5), (257, 64, 16, 0.3), (1000, 100, 32, 0.2),
                                      (5000, 333, 100, 0.5), (3000, 10, 3, 5.0), (8192, 999, 64, 0.05),
                                      (20000, 512, 64, 0.1), (20000, 64, 16, 1.0), (70001, 129, 32, 0.15), (140000, 64, 16, 0.1),
                                      (262144, 40, 8, 0.08)])
def test_pruned_uniform_cloud_matches_oracle(oracle, n, m, ns, r):
    rng = np.random.default_rng(n + m + ns)
    xyz = rng.uniform(-1, 1, size=(2, n, 3)).astype(np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, rng.integers(0, n, size=m)])
    new_xyz[0, 0] = [50, 50, 50]          # far outside: no hit -> zero row
    new_xyz[1, m - 1] += 0.01             # not a cloud point
    got = pruned(dev(new_xyz), dev(xyz), r, ns).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.ball_query(new_xyz, xyz, r, ns))


def test_pruned_anisotropic_and_clustered_clouds(oracle):
    rng = np.random.default_rng(7)
    n, m = 30000, 700
    flat = rng.uniform(-4, 4, size=(n, 3)).astype(np.float32)
    flat[:, 2] = (0.001 * rng.standard_normal(n)).astype(np.float32)          # a floor: one cell thick
    blobs = (rng.integers(0, 5, size=(n, 1)) * 3.0 + 0.05 * rng.standard_normal((n, 3))).astype(np.float32)
    long_axis = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    long_axis[:, 0] *= 40.0                                                    # > 32 cells on x: stretched cells
    for cloud, r, ns in ((flat, 0.2, 64), (blobs, 0.1, 32), (blobs, 0.4, 16), (long_axis, 0.3, 16)):
        xyz = cloud[None]
        new_xyz = np.ascontiguousarray(xyz[:, rng.integers(0, n, size=m)])
        new_xyz[0, :50] = rng.uniform(-6, 6, size=(50, 3)).astype(np.float32)  # some beyond the bounding box
        got = pruned(dev(new_xyz), dev(xyz), r, ns).cpu().numpy()
        np.testing.assert_array_equal(got, oracle.ball_query(new_xyz, xyz, r, ns))


def test_pruned_non_finite_and_degenerate_inputs(oracle):
    rng = np.random.default_rng(11)
    n, m = 9000, 257                       # b*m odd: no XCD remap
    xyz = rng.uniform(-1, 1, size=(1, n, 3)).astype(np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, rng.integers(0, n, size=m)])
    bad = xyz.copy()
    bad[0, 17] = np.nan
    bad[0, 4000, 1] = np.inf
    bad[0, 8000, 2] = -np.inf
    ctr = new_xyz.copy()
    ctr[0, 3] = np.nan
    ctr[0, 5, 0] = np.inf
    same = np.repeat(xyz[:, :1], n, axis=1)  # every point identical: zero extent
    with np.errstate(invalid="ignore"):
        cases = [(bad, ctr, 0.3, 16), (same, np.ascontiguousarray(same[:, :m]), 0.1, 8), (xyz, new_xyz, 0.0, 8),
                 (xyz, new_xyz, -0.25, 8), (xyz, new_xyz, 1e-30, 4), (xyz, new_xyz, 1e20, 4),
                 (xyz, new_xyz, float("inf"), 4), (xyz, new_xyz, float("nan"), 4)]
        for pts, centres, r, ns in cases:
            got = pruned(dev(centres), dev(pts), r, ns).cpu().numpy()
            np.testing.assert_array_equal(got, oracle.ball_query(centres, pts, r, ns), err_msg=f"r={r}")


def test_pruned_equals_streaming_at_north_star_shape():
    """SA1 of BASELINE.json configs[1]: 8 scenes x 50 000 points, 2048 FPS centres, r = 0.2, 64 samples --
    and the public wrapper routes this shape through the pruned path."""
    from butd_detr_amd import _hiplib, pointnet2_ext as ext
    pcs = dev(np.ascontiguousarray(scene_batch(8, 1184, 50000)[..., :3]))
    inds = ext.furthest_point_sampling(pcs, 2048)
    new_xyz = torch.gather(pcs, 1, inds.long()[..., None].expand(-1, -1, 3)).contiguous()
    want = streaming(new_xyz, pcs, 0.2, 64)
    assert torch.equal(pruned(new_xyz, pcs, 0.2, 64), want)
    assert _hiplib.load().butd_ball_query_workspace_bytes(8, 50000, 2048) > 0
    assert torch.equal(ext.ball_query(new_xyz, pcs, 0.2, 64), want)
    # dense balls (hits >> nsample) and a radius that covers the whole scene
    for r, ns in ((0.8, 16), (30.0, 8)):
        assert torch.equal(pruned(new_xyz[:, :256].contiguous(), pcs, r, ns),
                           streaming(new_xyz[:, :256].contiguous(), pcs, r, ns))


def test_ws_variant_falls_back_without_workspace(oracle):
    from butd_detr_amd import _hiplib
    lib = _hiplib.load()
    rng = np.random.default_rng(3)
    xyz = rng.uniform(-1, 1, size=(1, 2000, 3)).astype(np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, :100])
    assert lib.butd_ball_query_workspace_bytes(1, 2000, 100) == 0
    idx = torch.empty((1, 100, 8), dtype=torch.int32, device="cuda")
    a, b = dev(new_xyz), dev(xyz)
    assert lib.butd_ball_query_ws(1, 2000, 100, 0.3, 8, a.data_ptr(), b.data_ptr(), idx.data_ptr(), None, 0,
                                  torch.cuda.current_stream().cuda_stream) == 0
    np.testing.assert_array_equal(idx.cpu().numpy(), oracle.ball_query(new_xyz, xyz, 0.3, 8))
