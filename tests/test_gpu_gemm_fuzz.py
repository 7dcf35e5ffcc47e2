"""-m gpu: butd_gemm_grouped on randomly drawn problems -- all three operand layouts (forward, data
gradient, weight gradient), ragged sizes, bias / ReLU / scale epilogues, per-channel affine prologues,
split-K accumulation with the ones-column bias gradient, c_add / c2 accumulation, column statistics --
against float64 torch.matmul.  Sizes straddle the tile-configuration thresholds (32x32 vs 64x64 tiles,
fast vs generic instantiation)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(a, b, tol=2e-4, what=""):
    a, b = a.double().cpu().numpy(), b.double().cpu().numpy()
    scale = max(np.abs(b).max(), 1e-6)
    err = np.abs(a - b).max() / scale
    assert err <= tol, f"{what}: {err:.3e}"


TILES = [(0, 0), (32, 32), (32, -32), (64, 64), (64, -64), (32, 96), (32, -96), (64, 96), (64, -96), (96, 32), (96, -32),
         (128, 64), (128, -64), (128, 96), (128, -96)]   # negative: the one-slab-ahead K loop


@pytest.fixture(params=TILES, ids=lambda t: "auto" if t == (0, 0) else "tile%dx%d%s" % (t[0], abs(t[1]), "-one-ahead" if t[1] < 0 else ""))
def tile(request):
    """Every workgroup tile of the kernel's menu (butd_gemm_set_tile), and the built-in choice."""
    from butd_detr_amd import _hiplib
    lib = _hiplib.load()
    assert lib.butd_gemm_set_tile(*request.param) == 0
    yield request.param
    assert lib.butd_gemm_set_tile(0, 0) == 0


def test_set_tile_rejects_unknown():
    from butd_detr_amd import _hiplib
    lib = _hiplib.load()
    assert lib.butd_gemm_set_tile(48, 48) != 0
    assert lib.butd_gemm_set_tile(0, 0) == 0


@pytest.fixture(params=["f32", "bf16"])
def dtype(request):
    """'bf16': operands rounded to bf16 on the matrix cores (compute_bf16), fp32 accumulation -- checked against
    float64 products of the bf16-ROUNDED operands, i.e. the rounding is the only difference allowed."""
    from butd_detr_amd import fused_attention as fa
    prev = fa.set_compute_dtype(request.param)
    yield request.param
    fa.set_compute_dtype(prev)


def _close_any(a, refs, tol, what=""):
    """bf16 mode: a problem the float4-streaming kernel cannot take (odd sizes) is multiplied in fp32, the
    others on bf16-rounded operands -- accept whichever reference applies."""
    errs = []
    for b in refs:
        x, y = a.double().cpu().numpy(), b.double().cpu().numpy()
        errs.append(np.abs(x - y).max() / max(np.abs(y).max(), 1e-6))
    assert min(errs) <= tol, f"{what}: {min(errs):.3e}"


def _op(t, dtype):
    """An operand as the kernel multiplies it: fp32, or rounded to bf16 (nearest even)."""
    return t.double() if dtype == "f32" else t.bfloat16().double()


@pytest.mark.parametrize("seed", range(12))
def test_random_group(seed, tile, dtype):
    from butd_detr_amd import fused_attention as fa
    # bf16: a prologue affine is evaluated in fp32 and THEN rounded; an fma-vs-mul+add difference of one fp32
    # ulp there can move the bf16 rounding of an operand element by one bf16 ulp (2^-8 relative)
    tol_scale = 1.0 if dtype == "f32" else 20.0
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(seed)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    probs, checks, descr, keep = [], [], [], []   # keep: the problems hold raw pointers, not references
    for _ in range(int(rng.integers(1, 5))):
        kind = rng.choice(["fwd", "dgrad", "wgrad"])
        M = int(rng.choice([1, 7, 33, 64, 100, 257, 640, 2048, 5000]))
        N = int(rng.choice([3, 4, 32, 64, 100, 132, 288]))
        K = int(rng.choice([3, 6, 8, 36, 64, 131, 132, 288]))
        descr.append((str(kind), M, N, K))
        if kind == "fwd":
            x, w, y = r(M, K), r(N, K), torch.full((M, N), float("nan"), device=dev)
            bias = r(N) if rng.random() < 0.5 else None
            relu = bool(rng.random() < 0.3)
            scale = float(rng.choice([1.0, 0.5]))
            aff = (torch.rand(K, device=dev, generator=g) + 0.5, r(K)) if rng.random() < 0.3 else None
            c_add = bool(rng.random() < 0.3)
            c2 = r(M, N) if rng.random() < 0.3 else None
            stats = torch.zeros(2, N, dtype=torch.float64, device=dev) if (rng.random() < 0.3) else None
            if c_add:
                y = r(M, N)
            y0, c20 = y.clone(), None if c2 is None else c2.clone()
            keep += [x, w, y, bias, aff, c2, stats]
            probs.append(fa._fwd(x, w, y, M, N, K, bias=bias, relu=relu, scale=scale, a_affine=aff, c_add=c_add,
                                 c2=c2, col_stats=None if stats is None else (stats[0], stats[1])))
            xa = x if aff is None else torch.relu(x * aff[0] + aff[1])

            def finish(ref, bias=bias, scale=scale, relu=relu):
                if bias is not None:
                    ref = ref + bias.double()
                ref = ref * scale
                return torch.relu(ref) if relu else ref
            ref = finish(_op(xa, dtype) @ _op(w, dtype).t())
            ref32 = finish(xa.double() @ w.double().t())

            def check(y=y, refs=(ref, ref32), c_add=c_add, y0=y0, c2=c2, c20=c20, stats=stats):
                _close_any(y, [r + y0.double() if c_add else r for r in refs], 2e-4 * tol_scale)
                if c2 is not None:
                    _close_any(c2, [c20.double() + r for r in refs], 2e-4 * tol_scale)
                if stats is not None:
                    _close_any(stats[0], [r.sum(0) for r in refs], 1e-3 * tol_scale)
                    _close_any(stats[1], [(r * r).sum(0) for r in refs], 1e-3 * tol_scale)
            checks.append(check)
        elif kind == "dgrad":
            dy, w, dx = r(M, N), r(N, K), torch.full((M, K), float("nan"), device=dev)
            # the ReLU / dropout gate of the gradient this product creates, applied in its epilogue (c_gate)
            gate = torch.relu(r(M, K)) if rng.random() < 0.4 else None
            gscale = float(rng.choice([1.0, 1.25]))
            keep += [dy, w, dx, gate]
            probs.append(fa._dgrad(dy, w, dx, M, N, K, gate=gate, gate_scale=gscale))

            def check(dx=dx, dy=dy, w=w, gate=gate, gscale=gscale):
                refs = [_op(dy, dtype) @ _op(w, dtype), dy.double() @ w.double()]
                if gate is not None:
                    refs = [torch.where(gate > 0, ref * gscale, torch.zeros_like(ref)) for ref in refs]
                _close_any(dx, refs, 2e-4)
            checks.append(check)
        else:
            dy, x = r(M, N), r(M, K)
            dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
            with_bias = bool(rng.random() < 0.6)
            # BatchNorm + ReLU of the producing layer folded into the B operand (channel = column of x)
            baff = (torch.rand(K, device=dev, generator=g) + 0.5, r(K)) if rng.random() < 0.3 else None
            keep += [dy, x, dw, db, baff]
            probs.append(fa._wgrad(dy, x, dw, db if with_bias else None, M, N, K, b_affine=baff))
            xe = x if baff is None else torch.relu(x * baff[0] + baff[1])

            def check(dw=dw, db=db, dy=dy, xe=xe, with_bias=with_bias):
                _close_any(dw, [_op(dy, dtype).t() @ _op(xe, dtype), dy.double().t() @ xe.double()], 5e-4 * tol_scale)
                if with_bias:
                    _close_any(db, [_op(dy, dtype).sum(0), dy.double().sum(0)], 5e-4)
            checks.append(check)
    fa._gemm(probs, torch.empty(1, device=dev))
    torch.cuda.synchronize()
    for c, d in zip(checks, descr):
        try:
            c()
        except AssertionError as e:
            raise AssertionError(f"{d} (group {descr}): {e}") from None
