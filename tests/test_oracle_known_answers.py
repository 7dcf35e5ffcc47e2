"""Pins oracle/pointnet2_oracle.c: (1) against the literal CUDA-thread emulation, (2) against
hand-computable known answers for every rule SURVEY.md section 8(c) lists (tie rule, origin skip,
ball with 0/1/<ns/>ns hits, d2 == r^2 boundary)."""
import numpy as np
import pytest

from oracle import cuda_thread_emulation as emu


def bitrev(v, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (v & 1)
        v >>= 1
    return r


def test_opt_n_threads_table(oracle):
    for w in [1, 2, 3, 4, 7, 8, 9, 15, 16, 31, 32, 63, 64, 100, 127, 128, 132, 255, 256, 257,
              511, 512, 513, 1000, 1024, 2048, 4096, 50000, 200000]:
        assert oracle.opt_n_threads(w) == emu.opt_n_threads(w)
    assert oracle.opt_n_threads(50000) == 512
    assert oracle.opt_n_threads(256) == 256
    assert oracle.opt_n_threads(132) == 128


@pytest.mark.parametrize("n,m,seed", [(37, 20, 0), (64, 64, 1), (300, 50, 2), (700, 64, 3),
                                      (1500, 40, 4)])
def test_fps_matches_thread_emulation(oracle, n, m, seed):
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-1, 1, size=(2, n, 3)).astype(np.float32)
    pts[0, 5] = pts[0, 3]          # exact duplicates -> ties
    pts[1, n // 2] = 0.0           # origin point -> skip rule
    pts[1, n // 3] = [0.01, 0.01, 0.01]  # |p|^2 = 3e-4 <= 1e-3 -> skipped too
    ref = emu.furthest_point_sampling(pts, m)
    got = oracle.furthest_point_sampling(pts, m)
    np.testing.assert_array_equal(got, ref)
    got_mt = oracle.furthest_point_sampling(pts, m, multithread=True)
    np.testing.assert_array_equal(got_mt, ref)


def test_fps_mt_matches_single_large(oracle):
    from butd_detr_amd.synthetic_scenes import scannet_like_scene
    pc = scannet_like_scene(1184, 20000)[None, :, :3]
    a = oracle.furthest_point_sampling(pc, 128)
    b = oracle.furthest_point_sampling(pc, 128, multithread=True)
    np.testing.assert_array_equal(a, b)


def test_fps_collinear_known_answer(oracle):
    # points on a line at x = 1..9 (offset so none is near the origin); index 0 is x=1.
    xs = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9], dtype=np.float32)
    pts = np.zeros((1, 9, 3), dtype=np.float32)
    pts[0, :, 0] = xs
    pts[0, :, 1] = 1.0
    # FPS by hand: start 0 (x=1); farthest x=9 (idx 8); then x=5 (idx 4, d=16 from both);
    # then min-dists: x=3 ->4, x=7 ->4 (tie), x=2,4,6,8 -> 1. block_size = 8, slots = k % 8:
    # idx 2 -> slot 2 (bitrev3 = 2), idx 6 -> slot 6 (bitrev3 = 3): idx 2 wins.  Then idx 6.
    got = oracle.furthest_point_sampling(pts, 5)[0]
    assert got.tolist() == [0, 8, 4, 2, 6]


def test_fps_tie_rule_is_bit_reversed_slot(oracle):
    # n = 1024 -> block_size 512.  All points identical except index 0 -> after the first pick every
    # other point has the same distance: the winner must be the smallest bit-reversed (k % 512),
    # then smallest k within the slot.
    n = 1024
    pts = np.ones((1, n, 3), dtype=np.float32)
    pts[0, 0] = [3.0, 1.0, 1.0]
    got = oracle.furthest_point_sampling(pts, 2)[0]
    # slot 0 holds k=0 (dist 0 to itself) and k=512 (dist 4): slot 0's best is k=512, bitrev(0)=0 wins.
    assert got.tolist() == [0, 512]
    # make slot 0 lose: move k=512 onto the seed point; next best slot by bit reversal is 256 (bitrev=1)
    pts[0, 512] = pts[0, 0]
    got = oracle.furthest_point_sampling(pts, 2)[0]
    assert got.tolist() == [0, 256]
    order = sorted(range(512), key=lambda s: bitrev(s, 9))
    assert order[:4] == [0, 256, 128, 384]


def test_fps_origin_skip_and_index0_always_first(oracle):
    pts = np.array([[[0, 0, 0], [0.02, 0.0, 0.0], [1, 0, 0], [2, 0, 0], [0, 0.031, 0]]],
                   dtype=np.float32)
    # idx 0 is the origin but is always the first sample; idx 1 (|p|^2=4e-4) and idx 4
    # (|p|^2=9.61e-4) are skipped forever; 0.0317^2 > 1e-3 would not be.
    got = oracle.furthest_point_sampling(pts, 4)[0]
    assert got.tolist()[:3] == [0, 3, 2]
    # all remaining candidates are skipped or exhausted: slot bests of (-1, 0)/(0-dist) -> defined
    assert got[3] in (2, 3)
    ref = emu.furthest_point_sampling(pts, 4)[0]
    assert got.tolist() == ref.tolist()


def test_fps_mag_threshold_is_double_compare(oracle):
    # mag == float32(1e-3) is NOT <= 1e-3 (double) because float32(1e-3) > 1e-3.
    v = np.float32(1e-3)
    assert float(v) > 1e-3
    x = np.sqrt(np.float64(v)).astype(np.float32)
    # find an x whose fp32 square is exactly float32(1e-3) or just above/below
    cands = [np.nextafter(x, np.float32(0)), x, np.nextafter(x, np.float32(1))]
    for c in cands:
        mag = np.float32(c * c)
        pts = np.array([[[5, 5, 5], [c, 0, 0], [5, 5, 5.5]]], dtype=np.float32)
        got = oracle.furthest_point_sampling(pts, 2)[0]
        expect = 2 if float(mag) <= 1e-3 else 1
        assert got[1] == expect, (c, mag)


@pytest.mark.parametrize("n,m,ns,r,seed", [(200, 30, 8, 0.4, 0), (333, 17, 16, 0.25, 1),
                                           (64, 64, 4, 0.7, 2)])
def test_ball_query_matches_thread_emulation(oracle, n, m, ns, r, seed):
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-1, 1, size=(2, n, 3)).astype(np.float32)
    new_xyz = xyz[:, rng.permutation(n)[:m]].copy()
    new_xyz[0, 0] = [9, 9, 9]  # no hit at all
    ref = emu.ball_query(new_xyz, xyz, r, ns)
    got = oracle.ball_query(new_xyz, xyz, r, ns)
    np.testing.assert_array_equal(got, ref)


def test_ball_query_known_answers(oracle):
    xyz = np.zeros((1, 8, 3), dtype=np.float32)
    xyz[0, :, 0] = [0.0, 0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 10.0]
    centres = np.array([[[100, 0, 0],     # 0 hits  -> all zeros
                         [10.0, 0, 0],    # 1 hit   -> padded with 7
                         [1.0, 0, 0],     # r=1: |dx|<1 -> idx 1,2,3 ; 0 and 4 are at exactly r (strict <)
                         [1.5, 0, 0]]],   # r=1: idx 2,3,4 -> first 2 only when nsample=2
                       dtype=np.float32)
    got = oracle.ball_query(centres, xyz, 1.0, 4)[0]
    assert got[0].tolist() == [0, 0, 0, 0]
    assert got[1].tolist() == [7, 7, 7, 7]
    assert got[2].tolist() == [1, 2, 3, 1]
    assert got[3].tolist() == [2, 3, 4, 2]
    got2 = oracle.ball_query(centres, xyz, 1.0, 2)[0]
    assert got2[2].tolist() == [1, 2]
    assert got2[3].tolist() == [2, 3]


def test_ball_query_radius2_is_fp32_product(oracle):
    # radius2 = radius*radius rounded to fp32 (ball_query_gpu.cu:27); a point at d2 == radius2 is out.
    r = np.float32(0.2)
    r2 = np.float32(r * r)
    d = np.sqrt(np.float64(r2))
    xyz = np.zeros((1, 3, 3), dtype=np.float32)
    xyz[0, 1, 0] = np.float32(d)
    xyz[0, 2, 0] = np.nextafter(np.float32(d), np.float32(0))
    c = np.zeros((1, 1, 3), dtype=np.float32)
    got = oracle.ball_query(c, xyz, float(r), 3)[0, 0]
    exp = [0]
    for k in (1, 2):
        if np.float32(xyz[0, k, 0] * xyz[0, k, 0]) < r2:
            exp.append(k)
    exp = exp + [exp[0]] * (3 - len(exp))
    assert got.tolist() == exp


@pytest.mark.parametrize("n,m,seed", [(50, 20, 0), (33, 3, 1), (10, 2, 2)])
def test_three_nn_matches_thread_emulation(oracle, n, m, seed):
    rng = np.random.default_rng(seed)
    unknown = rng.uniform(-1, 1, size=(2, n, 3)).astype(np.float32)
    known = rng.uniform(-1, 1, size=(2, m, 3)).astype(np.float32)
    if m >= 3:
        known[0, 2] = known[0, 0]  # tie -> lower k first
    d_ref, i_ref = emu.three_nn(unknown, known)
    d_got, i_got = oracle.three_nn(unknown, known)
    np.testing.assert_array_equal(i_got, i_ref)
    np.testing.assert_array_equal(d_got, d_ref)
    if m < 3:  # unfilled slots keep 1e40 -> +inf in fp32, index 0
        assert np.isinf(d_got[..., 2]).all() and (i_got[..., 2] == 0).all()


def test_gather_group_interpolate_against_numpy(oracle):
    rng = np.random.default_rng(0)
    b, c, n, m, s = 2, 5, 40, 7, 3
    pts = rng.normal(size=(b, c, n)).astype(np.float32)
    idx = rng.integers(0, n, size=(b, m)).astype(np.int32)
    out = oracle.gather_points(pts, idx)
    np.testing.assert_array_equal(out, np.take_along_axis(pts, idx[:, None, :].repeat(c, 1), 2))
    gidx = rng.integers(0, n, size=(b, m, s)).astype(np.int32)
    gout = oracle.group_points(pts, gidx)
    for bi in range(b):
        np.testing.assert_array_equal(gout[bi], pts[bi][:, gidx[bi]])
    g = rng.normal(size=(b, c, m, s)).astype(np.float32)
    gg = oracle.group_points_grad(g, gidx, n)
    ref = np.zeros((b, c, n), dtype=np.float64)
    for bi in range(b):
        for j in range(m):
            for k in range(s):
                ref[bi, :, gidx[bi, j, k]] += g[bi, :, j, k]
    np.testing.assert_allclose(gg, ref, rtol=1e-5, atol=1e-6)
    g1 = rng.normal(size=(b, c, m)).astype(np.float32)
    gp = oracle.gather_points_grad(g1, idx, n)
    ref = np.zeros((b, c, n), dtype=np.float64)
    for bi in range(b):
        for j in range(m):
            ref[bi, :, idx[bi, j]] += g1[bi, :, j]
    np.testing.assert_allclose(gp, ref, rtol=1e-5, atol=1e-6)
    # three_interpolate: (p1*w1 + p2*w2) + p3*w3 in fp32
    idx3 = rng.integers(0, n, size=(b, m, 3)).astype(np.int32)
    w = rng.random((b, m, 3)).astype(np.float32)
    out = oracle.three_interpolate(pts, idx3, w)
    for bi in range(b):
        p = pts[bi][:, idx3[bi]]  # c, m, 3
        exp = (p[..., 0] * w[bi, :, 0] + p[..., 1] * w[bi, :, 1]) + p[..., 2] * w[bi, :, 2]
        np.testing.assert_array_equal(out[bi], exp.astype(np.float32))


def test_three_interpolate_grad_reference_case(oracle):
    """The one case the reference itself tests (pointnet2/pointnet2_test.py:18-30): feats (1,2,4),
    idx [[0,1,2],[1,2,3]], weight [[1,1,1],[2,2,2]] -- analytic Jacobian instead of gradcheck."""
    idx = np.array([[[0, 1, 2], [1, 2, 3]]], dtype=np.int32)
    w = np.array([[[1, 1, 1], [2, 2, 2]]], dtype=np.float32)
    g = np.array([[[1.0, 10.0], [100.0, 1000.0]]], dtype=np.float32)  # (1, 2, n=2)
    got = oracle.three_interpolate_grad(g, idx, w, 4)
    exp = np.array([[[1, 1 + 20, 1 + 20, 20], [100, 100 + 2000, 100 + 2000, 2000]]], dtype=np.float32)
    np.testing.assert_array_equal(got, exp)
    feats = np.arange(8, dtype=np.float32).reshape(1, 2, 4)
    out = oracle.three_interpolate(feats, idx, w)
    np.testing.assert_array_equal(out, np.array([[[3, 12], [15, 36]]], dtype=np.float32))
