"""The compiled ATen / pybind11 front end of the C ABI (butd_detr_amd/binding/pointnet2_aten.cpp): the nine
functions of the reference's `pointnet2._ext` (pointnet2/_ext_src/src/bindings.cpp:11-24) with its argument
checks -- the binding a maintainer keeps when the reference's own extension build stays in place
(INTEGRATION.md section 1).  CPU part: it compiles against include/butd_pointnet2.h + torch's headers, links
libbutd_detr_hip.so, exports the module surface and rejects what the reference rejects.  GPU part: the same
results, bit for bit, as the ctypes binding the product path uses."""
import numpy as np
import pytest
import torch

EXPORTS = ("gather_points", "gather_points_grad", "furthest_point_sampling", "three_nn", "three_interpolate",
           "three_interpolate_grad", "ball_query", "group_points", "group_points_grad")


@pytest.fixture(scope="module")
def ext():
    from butd_detr_amd.binding import build
    return build.load()


def test_builds_and_exports_the_reference_module_surface(ext):
    for name in EXPORTS:
        assert callable(getattr(ext, name)), name


def test_rejects_what_the_reference_rejects(ext):
    pts = torch.rand(1, 64, 3)
    with pytest.raises(RuntimeError, match="CPU not supported"):          # sampling.cpp:86-88
        ext.furthest_point_sampling(pts, 8)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        ext.ball_query(pts[:, :4], pts, 0.2, 4)
    with pytest.raises(RuntimeError, match="contiguous"):                 # utils.h:16-19
        ext.furthest_point_sampling(torch.rand(1, 3, 64).transpose(1, 2), 8)
    with pytest.raises(RuntimeError, match="float"):                      # utils.h:26-30
        ext.furthest_point_sampling(pts.double(), 8)
    with pytest.raises(RuntimeError, match="int"):                        # utils.h:21-24
        ext.gather_points(torch.rand(1, 4, 64), torch.zeros(1, 8, dtype=torch.int64))


@pytest.mark.gpu
def test_same_results_as_the_ctypes_binding(ext):
    from butd_detr_amd import pointnet2_ext as ct
    g = torch.Generator(device="cuda").manual_seed(0)
    b, n, m, c = 2, 9000, 512, 16
    xyz = torch.rand(b, n, 3, device="cuda", generator=g) * 4 - 2
    feats = torch.randn(b, c, n, device="cuda", generator=g)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                                         # launches follow the CURRENT stream
        inds = ext.furthest_point_sampling(xyz, m)
        new_xyz = ext.gather_points(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
        idx = ext.ball_query(new_xyz, xyz, 0.3, 16)
        grouped = ext.group_points(feats, idx)
        d2, i3 = ext.three_nn(xyz, new_xyz)
        w = torch.softmax(-d2, -1)
        interp = ext.three_interpolate(grouped[..., 0].contiguous(), i3, w)
        g_interp = ext.three_interpolate_grad(interp, i3, w, m)
        g_group = ext.group_points_grad(grouped, idx, n)
        g_gather = ext.gather_points_grad(new_xyz.transpose(1, 2).contiguous(), inds, n)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(inds, ct.furthest_point_sampling(xyz, m))
    assert torch.equal(idx, ct.ball_query(new_xyz, xyz, 0.3, 16))
    assert torch.equal(grouped, ct.group_points(feats, idx))
    d2_c, i3_c = ct.three_nn(xyz, new_xyz)
    assert torch.equal(d2, d2_c) and torch.equal(i3, i3_c)
    assert torch.equal(interp, ct.three_interpolate(grouped[..., 0].contiguous(), i3, w))
    for got, want in ((g_interp, ct.three_interpolate_grad(interp, i3, w, m)),
                      (g_group, ct.group_points_grad(grouped, idx, n)),
                      (g_gather, ct.gather_points_grad(new_xyz.transpose(1, 2).contiguous(), inds, n))):
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-5)   # atomics
    # the large-cloud paths (pruned FPS / grid ball query) are taken through the same entry points
    big = torch.rand(1, 60000, 3, device="cuda", generator=g) * 6
    inds_b = ext.furthest_point_sampling(big, 2048)
    assert torch.equal(inds_b, ct.furthest_point_sampling(big, 2048))
    cb = torch.gather(big, 1, inds_b.long()[..., None].expand(-1, -1, 3)).contiguous()
    assert torch.equal(ext.ball_query(cb, big, 0.2, 64), ct.ball_query(cb, big, 0.2, 64))


def test_dropin_can_register_it_as_the_reference_extension(ext, monkeypatch):
    import sys
    for k in [k for k in sys.modules if k.startswith("pointnet2")]:
        monkeypatch.delitem(sys.modules, k)
    from butd_detr_amd import dropin
    dropin.install(scope="ops", ops_binding="aten")
    import pointnet2._ext as _ext
    assert _ext is ext and callable(_ext.ball_query)
