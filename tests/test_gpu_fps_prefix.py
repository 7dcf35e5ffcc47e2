"""-m gpu: furthest point sampling of clouds that are already in furthest-point order (levels 2-4 of the backbone,
models/backbone_module.py:131-140).  The library proves ``inds == arange(m)`` with parallel work
(``butd_fps_prefix_check``: an exact test incl. the reference's tie order) and runs the serial kernel only where the
answer is no: the indices are the oracle's bit for bit either way (sampling_gpu.cu:74-178 restated in
oracle/pointnet2_oracle.c), and the verdict is 0 exactly where the oracle's indices are 0..m-1."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from butd_detr_amd.synthetic_scenes import scene_batch  # noqa: E402


@pytest.fixture(scope="module")
def ext():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from butd_detr_amd import pointnet2_ext
    return pointnet2_ext


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def ordered(oracle):
    """8 scenes x 2048 points in the order SA1's sampling selected them (the input of SA2)."""
    pcs = np.ascontiguousarray(scene_batch(8, 77, 20000)[..., :3])
    inds = oracle.furthest_point_sampling(pcs, 2048, multithread=True)
    return np.ascontiguousarray(np.take_along_axis(pcs, inds[..., None].astype(np.int64), 1))


def both(ext, oracle, pts, m):
    verdict = ext.fps_prefix_verdict(dev(pts), m)
    got = ext.furthest_point_sampling(dev(pts), m).cpu().numpy()
    ref = oracle.furthest_point_sampling(pts, m)
    np.testing.assert_array_equal(got, ref)
    if verdict is None:
        return None, got
    verdict = verdict.cpu().numpy()
    np.testing.assert_array_equal(verdict == 0, (ref == np.arange(m, dtype=ref.dtype)).all(axis=1))   # exact, both ways
    return verdict, got


@pytest.mark.parametrize("n,m", [(2048, 1024), (1024, 512), (512, 256), (2048, 2048), (300, 7), (2048, 2)])
def test_ordered_levels_are_proven_and_exact(ext, oracle, ordered, n, m):
    verdict, got = both(ext, oracle, ordered[:, :n], m)
    assert (verdict == 0).all()
    np.testing.assert_array_equal(got, np.broadcast_to(np.arange(m, dtype=np.int32), got.shape))


def test_largest_covered_shape(ext, oracle):
    pcs = np.random.default_rng(11).uniform(-3, 3, size=(2, 20000, 3)).astype(np.float32)   # (no exact ties)
    inds = oracle.furthest_point_sampling(pcs, 8192, multithread=True)
    pts = np.ascontiguousarray(np.take_along_axis(pcs, inds[..., None].astype(np.int64), 1))
    verdict, got = both(ext, oracle, pts, 2048)
    assert (verdict == 0).all() and (got == np.arange(2048)).all()
    # scenes with boxes / planes hold exact ties: the tie order decides them inside the check (asserted in both())
    scenes = np.ascontiguousarray(scene_batch(2, 5, 20000)[..., :3])
    inds = oracle.furthest_point_sampling(scenes, 8192, multithread=True)
    spts = np.ascontiguousarray(np.take_along_axis(scenes, inds[..., None].astype(np.int64), 1))
    both(ext, oracle, spts, 2048)
    assert ext.fps_prefix_verdict(dev(pcs[:, :8193]), 2048) is None          # not covered: serial / pruned path only
    assert ext.fps_prefix_verdict(dev(pts), 2049) is None


def test_unordered_clouds_fall_back(ext, oracle):
    rng = np.random.default_rng(3)
    pts = rng.uniform(-2, 2, size=(4, 2048, 3)).astype(np.float32)
    verdict, _ = both(ext, oracle, pts, 1024)
    assert (verdict != 0).all()


def test_mixed_batch_per_scene_verdict(ext, oracle, ordered):
    pts = ordered[:4].copy()
    pts[1] = pts[1, np.random.default_rng(0).permutation(2048)]
    pts[3, [700, 701]] = pts[3, [701, 700]]                    # two neighbours of the prefix swapped
    verdict, got = both(ext, oracle, pts, 1024)
    assert verdict[0] == 0 and verdict[2] == 0 and verdict[1] != 0 and verdict[3] != 0
    assert (got[0] == np.arange(1024)).all() and got[3, 700] == 701 and got[3, 701] == 700


def test_ties_and_skipped_points(ext, oracle, ordered):
    # (a) exact copies of prefix points far behind the prefix tie with them at their iteration: the reference's tie
    #     order (bit-reversed slot, fps_common.h) decides -- inside the check as in the serial kernel.  Over 24 copies
    #     both outcomes occur.
    outcomes = set()
    for j, k in zip(range(290, 314), range(1900, 1924)):
        a = ordered[0:1].copy()
        a[0, k] = a[0, j]
        verdict, _ = both(ext, oracle, a, 1024)
        outcomes.add(int(verdict[0] != 0))
    assert outcomes == {0, 1}
    # (b) a duplicate INSIDE the prefix: its threshold is 0 and it ties with every sample already taken
    b = ordered[1:2].copy()
    b[0, 301] = b[0, 300]
    both(ext, oracle, b, 1024)
    b[0, 1:40] = b[0, 0]
    both(ext, oracle, b, 1024)
    # (c) a prefix point inside the |p|^2 <= 1e-3 skip can never be selected
    c = ordered[2:3].copy()
    c[0, 17] = [0.01, 0.01, 0.01]
    verdict, got = both(ext, oracle, c, 1024)
    assert verdict[0] != 0 and 17 not in got[0]
    # (d) a skipped point BEHIND the prefix does not compete: still proven
    d = ordered[3:4].copy()
    d[0, 2000] = [0.0, 0.02, 0.0]
    verdict, got = both(ext, oracle, d, 1024)
    assert verdict[0] == 0 and (got[0] == np.arange(1024)).all()
    # (e) point 0 is the first sample whether skipped or not (sampling_gpu.cu:96-98)
    e = ordered[4:5].copy()
    e[0, 0] = [0.0, 0.0, 0.01]
    both(ext, oracle, e, 1024)


def test_backbone_chain_uses_the_proof(ext, oracle):
    """The four levels as the backbone calls them: level 1 on the raw cloud, levels 2-4 on gathered samples."""
    pcs = np.ascontiguousarray(scene_batch(2, 1184, 50000)[..., :3])
    xyz = dev(pcs)
    ref_xyz = pcs
    for level, m in enumerate((2048, 1024, 512, 256)):
        ref = oracle.furthest_point_sampling(ref_xyz, m, multithread=True)
        got = ext.furthest_point_sampling(xyz, m)
        np.testing.assert_array_equal(got.cpu().numpy(), ref)
        if level:
            assert (ext.fps_prefix_verdict(xyz, m) == 0).all()       # (scene 0 holds an exact tie at sample 826 of level 2)
        xyz = torch.gather(xyz, 1, got.long()[..., None].expand(-1, -1, 3)).contiguous()
        ref_xyz = np.ascontiguousarray(np.take_along_axis(ref_xyz, ref[..., None].astype(np.int64), 1))


def test_levels_2_to_4_time(ext, ordered):
    """Not a bound, a record: the three ordered levels of a batch of 8, proven vs forced onto the serial kernel."""
    pts = dev(ordered)
    shuffled = pts[:, torch.randperm(2048, device="cuda")].contiguous()

    def chain(x):
        for n, m in ((2048, 1024), (1024, 512), (512, 256)):
            ext.furthest_point_sampling(x[:, :n].contiguous(), m)

    for name, x in (("ordered", pts), ("shuffled", shuffled)):
        chain(x)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(5):
            chain(x)
        t1.record()
        torch.cuda.synchronize()
        print(f"FPS levels 2-4, 8 scenes, {name}: {t0.elapsed_time(t1) / 5 * 1e3:.0f} us")
