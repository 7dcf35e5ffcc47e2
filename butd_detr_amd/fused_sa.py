"""Set-abstraction shared MLP on the gfx950 pipeline of include/butd_sa.h.

``sa_mlp_pool`` = QueryAndGroup's gather/normalise + 3 x [1x1 conv -> BatchNorm2d -> ReLU] + max-pool
over nsample (pointnet2_modules.py:243-257), forward AND backward, in training (batch statistics,
running-stat update) or eval (running statistics) mode.  Activations are position-major (P, C)
matrices; each 1x1 conv is one MFMA GEMM launch whose operand load applies the previous layer's
BatchNorm+ReLU, so no NCHW tensor, transpose, BN-apply or ReLU pass ever touches HBM.
"""
import os

import torch

from . import _hiplib, switches
from .fused_attention import _dgrad, _flush_folds, _fwd, _gemm, _pending_folds, _problem, _wgrad, fold_scope, zeros

_lib = _hiplib.load()


def _s(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _call(name, ref, *args):
    with torch.cuda.device(ref.device):
        err = getattr(_lib, name)(*args, _s(ref))
    _hiplib.check(err, name)


# Training-mode backward of the last layer + max-pool by linearity (include/butd_sa.h, butd_sa_last_bwd): no dense dZ3,
# half the matrix work, 3.25 -> 1.34 GB at SA1.  BUTD_AB=sa_last_bwd=0 keeps the dense path (A/B switch, and the path of
# widths / nsample the kernel has no instance for).
_LAST_LIN = [switches.flag("sa_last_bwd", True)]
_LAST_FWD = [switches.flag("sa_last_fwd", True)]      # ... and the forward that does not write Z3
_FIRST_LIN = [switches.flag("sa_first_bwd", True)]     # ... and the first layer's, where no input gradient is wanted
_MID_FIRST = [switches.flag("sa_mid_bwd", True)]       # ... with layer 2's backward in the same pass (64-wide levels)
# SA2-4 (128-wide layers, input gradient wanted): layer 2's backward in one pass that writes only layer 1's gated gradient
_MID_WIDE = [switches.flag("sa_mid_wide", True)]
# ... and SA1's forward never writing Z1 (butd_sa_first_two_fwd): layer 1's BatchNorm sums from the 8 x 8 moments of X, z1
# formed on the matrix cores by layer 2's kernel and, with the same instructions, by the backward.  23.64 -> 23.44 ms
# (4 x 60 steps), and the SA1 gradients move 4-5x CLOSER to a float64 run (closer than stock torch's):
# profiles/r04_sa_last_layer.txt.  (A first version that recomputed z1 with scalar code from the LDS X tile was neutral.)
_NO_Z1 = [switches.flag("sa_no_z1", True)]
# SA2-4 (levels with input features): the first layer by linearity of the grouping (csrc/sa_first_linear.hip, round 6): the
# C-wide product over the level's N points instead of its P grouped rows, Z1 as a gather + three multiply-adds; backward: one
# pass over (g1, Z1) along the inverted neighbour lists, products over N rows.  The grouped input X is never formed.
_FIRST_LINEAR = [switches.flag("sa_first_linear", True)]
_FUSED_EVAL = [True]     # inference: a whole level as one kernel (sa_fused_eval)
_scratch_sizes = {}
_sched = {}


def _sched_words(dev):
    """Two uint32 per (device, stream) for the kernels that hand their blocks out through a counter (zero between
    launches: the kernel's last workgroup resets them)."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    t = _sched.get(key)
    if t is None:
        t = _sched[key] = torch.zeros(2, dtype=torch.int32, device=dev)      # (not from the step arena: it persists)
    return t


def set_last_layer_linear(flag):
    prev = _LAST_LIN[0]
    _LAST_LIN[0] = bool(flag)
    return prev


_LAST_MIN_ROWS = [0]


def _last_lin_ok(training, ns, C2, C3, P=1 << 30):
    return bool(training and _LAST_LIN[0] and P >= _LAST_MIN_ROWS[0]
                and _lib.butd_sa_last_bwd_supported(int(ns), int(C2), int(C3)))


def _last_scratch(P, C2, C3):
    key = (P, C2, C3)
    if key not in _scratch_sizes:
        import ctypes
        nf, nd = ctypes.c_long(0), ctypes.c_long(0)
        _hiplib.check(_lib.butd_sa_last_bwd_scratch(P, C2, C3, ctypes.byref(nf), ctypes.byref(nd)),
                      "butd_sa_last_bwd_scratch")
        _scratch_sizes[key] = (nf.value, nd.value)
    return _scratch_sizes[key]


_FUSE_STATS = [switches.flag("sa_fuse_stats", True)]   # layer 1's gate + BatchNorm-backward sums in the epilogue of the product that creates dH1
_GATHER = [True]      # feature gradient as a gather over the inverted neighbour lists


def inverse_index(idx, N):
    """(start (B*N + 1,) int32, list (B*np*ns,) int32): for every input point the grouped rows that copy it
    (include/butd_sa.h, butd_sa_inverse_index).  Coordinates only: Pointnet2Backbone.plan prefetches it."""
    B, np_, ns = idx.shape
    dev = idx.device
    count = torch.zeros(B * N, dtype=torch.int32, device=dev)      # (not the step arena: this also runs on the prefetch stream)
    start = torch.empty(B * N + 1, dtype=torch.int32, device=dev)
    lst = torch.empty(B * np_ * ns, dtype=torch.int32, device=dev)
    _call("butd_sa_inverse_index", idx, B, N, np_, ns, idx.data_ptr(), count.data_ptr(), start.data_ptr(), lst.data_ptr())
    return start, lst


def _mid_scratch(P, C, Kp):
    key = ("mid", P, C, Kp)
    if key not in _scratch_sizes:
        import ctypes
        nf, nd = ctypes.c_long(0), ctypes.c_long(0)
        _hiplib.check(_lib.butd_sa_mid_first_bwd_scratch(P, C, Kp, ctypes.byref(nf), ctypes.byref(nd)),
                      "butd_sa_mid_first_bwd_scratch")
        _scratch_sizes[key] = (nf.value, nd.value)
    return _scratch_sizes[key]


def _wide_scratch(P, C):
    key = ("wide", P, C)
    if key not in _scratch_sizes:
        import ctypes
        nf, nd = ctypes.c_long(0), ctypes.c_long(0)
        _hiplib.check(_lib.butd_sa_mid_wide_bwd_scratch(P, C, ctypes.byref(nf), ctypes.byref(nd)),
                      "butd_sa_mid_wide_bwd_scratch")
        _scratch_sizes[key] = (nf.value, nd.value)
    return _scratch_sizes[key]


def _first_scratch(P, C1, Kp):
    key = ("first", P, C1, Kp)
    if key not in _scratch_sizes:
        import ctypes
        nf, nd = ctypes.c_long(0), ctypes.c_long(0)
        _hiplib.check(_lib.butd_sa_first_bwd_scratch(P, C1, Kp, ctypes.byref(nf), ctypes.byref(nd)),
                      "butd_sa_first_bwd_scratch")
        _scratch_sizes[key] = (nf.value, nd.value)
    return _scratch_sizes[key]


class _SAMlpPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, feats_pm, new_xyz, idx, radius, normalize, training, momentum,
                w1, g1, b1, rm1, rv1, nbt1, eps1,
                w2, g2, b2, rm2, rv2, nbt2, eps2,
                w3, g3, b3, rm3, rv3, nbt3, eps3,
                feat_ptr_offset, feat_stride, inv_start=None, inv_list=None):
        B, N, _ = xyz.shape
        np_, ns = idx.shape[1], idx.shape[2]
        C = 0 if feats_pm is None else (feats_pm.shape[-1] - feat_ptr_offset)
        Cin = 3 + C
        P = B * np_ * ns
        dev = xyz.device
        ws = [w.reshape(w.shape[0], -1) for w in (w1, w2, w3)]
        C1, C2, C3 = (w.shape[0] for w in ws)
        assert ws[0].shape[1] == Cin and ws[1].shape[1] == C1 and ws[2].shape[1] == C2
        # rows of 3+C floats (6, 131, 259) are not 16-byte aligned: pad the grouped input and the first
        # weight matrix to a multiple of 4 columns (zeros) so the first product takes the float4 path
        Kp = (Cin + 3) // 4 * 4
        # the first layer by linearity (levels with features whose rows are float4-aligned: SA2-4): no grouped input
        lin1 = bool(_FIRST_LINEAR[0] and feats_pm is not None and C >= 16 and C % 4 == 0 and feat_ptr_offset == 0
                    and feat_stride == C and _lib.butd_sa_first_linear_supported(int(C1)))
        if Kp != Cin and not lin1:
            ws[0] = torch.nn.functional.pad(ws[0], (0, Kp - Cin))
        fptr = None if feats_pm is None else feats_pm.data_ptr() + 4 * feat_ptr_offset
        if lin1:
            X = xyz.new_empty((0, Kp))
            w1_full = w1.reshape(C1, Cin)
            Wf = w1_full[:, 3:].contiguous()                    # (C1, C)
        else:
            X = torch.empty((P, Kp), device=dev)
            _call("butd_sa_group", xyz, B, N, np_, ns, C, xyz.data_ptr(), new_xyz.data_ptr(), fptr,
                  feat_stride, idx.data_ptr(), float(radius), int(bool(normalize)), X.data_ptr(), Kp)

        Cm = max(C1, C2, C3)
        # 16 private copies of the epilogue's column sums (folded by butd_sa_bn_finalize): a 65 536-row product
        # ends in 1 024 double atomics per column address otherwise (65536x128x128: 64.9 -> 51.3 us).  (256
        # copies were measured for the 10^6-row layers: the per-tile epilogue work costs more than the
        # separate chunked pass.)
        SLOTS = 16
        stats = zeros((3, SLOTS, 2, Cm), dtype=torch.float64, device=dev)
        aff = torch.empty((3, 4, max(C1, C2, C3)), device=dev)  # per layer: mean, rstd, scale, shift
        layers = ((g1, b1, rm1, rv1, nbt1, eps1), (g2, b2, rm2, rv2, nbt2, eps2),
                  (g3, b3, rm3, rv3, nbt3, eps3))
        Zs, inp, prev_aff = [], X, None
        G = B * np_
        zmax = zmin = amax = amin = None
        lin = _last_lin_ok(training, ns, C2, C3, P)  # the backward then never reads Z3: the forward does not write it
        # SA1 (no input gradient, 8 grouped columns, 64-wide layers): layer 1's BatchNorm sums from the moments of X, z1
        # formed on the fly by layer 2's kernel and again by the backward (butd_sa_mid_first_bwd): Z1 is never written
        no_z1 = (lin and Kp == 8 and C1 == 64 and C2 == 64 and _NO_Z1[0] and _MID_FIRST[0] and _FIRST_LIN[0]
                 and not (feats_pm is not None and feats_pm.requires_grad))
        for li, (Cl, w) in enumerate(zip((C1, C2, C3), ws)):
            last = li == 2
            if no_z1 and li < 2:
                if li == 0:
                    mom = zeros(72, dtype=torch.float64, device=dev)
                    _call("butd_sa_first_two_fwd", xyz, P, C1, Kp, X.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(),
                          mom.data_ptr(), stats[0, 0, 0].data_ptr(), stats[0, 0, 1].data_ptr(), None, None, None, None,
                          None, 0)
                    Z = X[:0]
                else:
                    Z = torch.empty((P, Cl), device=dev)
                    _call("butd_sa_first_two_fwd", xyz, P, C1, Kp, X.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), None,
                          None, None, aff[0, 2].data_ptr(), aff[0, 3].data_ptr(), Z.data_ptr(),
                          stats[1, 0, 0].data_ptr(), stats[1, 0, 1].data_ptr(), 1)
                g, b, rm, rv, nbt, eps = layers[li]
                _call("butd_sa_bn_finalize", xyz, Cl, P, stats[li, 0, 0].data_ptr(), stats[li, 0, 1].data_ptr(), 1, 2 * Cm,
                      g.data_ptr(), b.data_ptr(), float(eps), float(momentum), int(training), rm.data_ptr(), rv.data_ptr(),
                      _p(nbt), aff[li, 0].data_ptr(), aff[li, 1].data_ptr(), aff[li, 2].data_ptr(), aff[li, 3].data_ptr())
                prev_aff = (aff[li, 2], aff[li, 3])
                Zs.append(Z)
                inp = Z
                continue
            if li == 0 and lin1:
                Y = torch.empty((B * N, C1), device=dev)
                _gemm([_problem(feats_pm, Wf, Y, B * N, C1, C, (feat_stride, 1), (C, 1), C1)], xyz)
                Z = torch.empty((P, C1), device=dev)
                _call("butd_sa_first_linear_fwd", xyz, B, N, np_, ns, C1, xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(),
                      float(radius), int(bool(normalize)), Y.data_ptr(), w1_full.data_ptr(), Cin, Z.data_ptr(),
                      stats[0, 0, 0].data_ptr() if training else None, stats[0, 0, 1].data_ptr() if training else None,
                      SLOTS, 2 * Cm)
                g, b, rm, rv, nbt, eps = layers[0]
                _call("butd_sa_bn_finalize", xyz, Cl, P, stats[0, 0, 0].data_ptr(), stats[0, 0, 1].data_ptr(),
                      SLOTS if training else 1, 2 * Cm, g.data_ptr(), b.data_ptr(), float(eps), float(momentum), int(training),
                      rm.data_ptr(), rv.data_ptr(), _p(nbt), aff[0, 0].data_ptr(), aff[0, 1].data_ptr(),
                      aff[0, 2].data_ptr(), aff[0, 3].data_ptr())
                prev_aff = (aff[0, 2], aff[0, 3])
                Zs.append(Z)
                inp = Z
                continue
            if last:
                zmax = torch.empty((G, Cl), device=dev)
                zmin = torch.empty((G, Cl), device=dev)
                amax = torch.empty((G, Cl), dtype=torch.uint8, device=dev)
                amin = torch.empty((G, Cl), dtype=torch.uint8, device=dev)
            if last and lin and _LAST_FWD[0]:
                # product + BatchNorm sums + pooling extrema in one pass over Z2 (include/butd_sa.h, butd_sa_last_fwd)
                _call("butd_sa_last_fwd", xyz, B, np_, ns, C2, C3, inp.data_ptr(), prev_aff[0].data_ptr(),
                      prev_aff[1].data_ptr(), w.data_ptr(), stats[li, 0, 0].data_ptr(), stats[li, 0, 1].data_ptr(),
                      zmax.data_ptr(), zmin.data_ptr(), amax.data_ptr(), amin.data_ptr(), _sched_words(dev).data_ptr())
                Z, in_gemm_stats = inp[:0], False
            else:
                Z = torch.empty((P, Cl), device=dev)
                # BatchNorm sums straight from the GEMM epilogue while the row count is moderate (every tile
                # ends in 2 double atomics per column); the last layer's pass also takes the pooling extrema
                # (limit measured in the step, round 4: up to 131 072 rows 25.62 ms, 262 144 rows 25.40, 1 048 576 rows 25.51)
                in_gemm_stats = training and not last and P <= 262144
                thin = li == 0 and Kp == 8      # SA1: xyz + colour -> one HBM pass does the product AND the sums
                if thin:
                    _call("butd_sa_thin_conv", xyz, P, Cl, Kp, X.data_ptr(), Kp, w.data_ptr(), Z.data_ptr(),
                          stats[li, 0, 0].data_ptr() if training else None,
                          stats[li, 0, 1].data_ptr() if training else None)
                    in_gemm_stats = training
                else:
                    _gemm([_fwd(inp, w, Z, P, Cl, w.shape[1], a_affine=prev_aff,
                                col_stats=(stats[li, 0, 0], stats[li, 0, 1]) if in_gemm_stats else None,
                                col_slots=(SLOTS, 2 * Cm) if in_gemm_stats else (0, 0))], xyz)
                if last or (training and not in_gemm_stats):
                    _call("butd_sa_colstats", xyz, P, Cl, Z.data_ptr(), stats[li, 0, 0].data_ptr(),
                          stats[li, 0, 1].data_ptr(), ns if last else 0, _p(zmax), _p(zmin), _p(amax), _p(amin))
            g, b, rm, rv, nbt, eps = layers[li]
            _call("butd_sa_bn_finalize", xyz, Cl, P, stats[li, 0, 0].data_ptr(), stats[li, 0, 1].data_ptr(),
                  SLOTS if in_gemm_stats else 1, 2 * Cm, g.data_ptr(), b.data_ptr(), float(eps), float(momentum), int(training), rm.data_ptr(),
                  rv.data_ptr(), _p(nbt), aff[li, 0].data_ptr(), aff[li, 1].data_ptr(),
                  aff[li, 2].data_ptr(), aff[li, 3].data_ptr())
            prev_aff = (aff[li, 2], aff[li, 3])
            Zs.append(Z)
            inp = Z
        out_cm = torch.empty((B, C3, np_), device=dev)
        out_pm = torch.empty((B, np_, C3), device=dev)
        zsel = torch.empty((G, C3), device=dev)
        asel = torch.empty((G, C3), dtype=torch.uint8, device=dev)
        _call("butd_sa_pool_finalize", xyz, B, np_, C3, zmax.data_ptr(), zmin.data_ptr(), amax.data_ptr(),
              amin.data_ptr(), aff[2, 2].data_ptr(), aff[2, 3].data_ptr(), out_cm.data_ptr(),
              out_pm.data_ptr(), zsel.data_ptr(), asel.data_ptr())
        ctx.save_for_backward(X, Zs[0], Zs[1], Zs[2], idx, aff, zsel, asel, ws[0], ws[1], ws[2],
                              g1, g2, g3)
        ctx.cfg = (B, N, np_, ns, C, bool(training), feats_pm is not None and feats_pm.requires_grad,
                   w1.shape, w2.shape, w3.shape, Cin, lin)
        ctx.no_z1 = no_z1
        ctx.inv = (inv_start, inv_list) if inv_start is not None else None
        # (the linear first layer's backward re-reads the coordinates and the features instead of X)
        ctx.lin1 = (xyz, new_xyz, feats_pm, Wf, float(radius), bool(normalize), P) if lin1 else None
        return out_cm, out_pm

    @staticmethod
    @fold_scope
    def backward(ctx, d_cm, d_pm):
        X, Z1, Z2, Z3, idx, aff, zsel, asel, w1, w2, w3, g1, g2, g3 = ctx.saved_tensors
        B, N, np_, ns, C, training, need_dfeat, s1, s2, s3, Cin, lin = ctx.cfg
        dev = X.device
        P, Kp = X.shape                                 # Kp = Cin rounded up to a multiple of 4
        if ctx.lin1 is not None:
            P = ctx.lin1[6]
        C1, C2, C3 = w1.shape[0], w2.shape[0], w3.shape[0]
        # gradient of the pooled output, position-major (B, np, C) like everything else here
        if d_cm is None:
            d_out = d_pm.contiguous()
        elif d_pm is None:
            d_out = d_cm.transpose(1, 2).contiguous()
        else:
            d_out = d_pm + d_cm.transpose(1, 2)
        tr = int(training)
        S = zeros((3, 2, max(C1, C2, C3)), dtype=torch.float64, device=dev)
        dW = zeros(C3 * C2 + C2 * C1 + C1 * Kp, device=dev)
        dW3 = dW[:C3 * C2].view(C3, C2)
        dW2 = dW[C3 * C2:C3 * C2 + C2 * C1].view(C2, C1)
        dW1 = dW[C3 * C2 + C2 * C1:].view(C1, Kp)
        mean = lambda l: aff[l, 0]
        rstd = lambda l: aff[l, 1]
        scale = lambda l: aff[l, 2]
        shift = lambda l: aff[l, 3]
        # ---- layer 3: max-pool + ReLU + BN
        _call("butd_sa_pool_bwd_stats", X, B, np_, C3, d_out.data_ptr(), zsel.data_ptr(), scale(2).data_ptr(),
              shift(2).data_ptr(), mean(2).data_ptr(), rstd(2).data_ptr(), S[2, 0].data_ptr(),
              S[2, 1].data_ptr())
        dH2 = torch.empty((P, C2), device=dev)
        if lin:
            # dH2 (gated by layer 2's ReLU), dW3 and layer 2's BatchNorm sums straight from Z2 and the pooled gradient
            nf, nd = _last_scratch(P, C2, C3)
            ws_f = torch.empty(nf, device=dev)
            ws_d = torch.empty(nd, dtype=torch.float64, device=dev)
            _call("butd_sa_last_bwd", X, B, np_, ns, C2, C3, Z2.data_ptr(), scale(1).data_ptr(), shift(1).data_ptr(),
                  mean(1).data_ptr(), rstd(1).data_ptr(), w3.data_ptr(), d_out.data_ptr(), zsel.data_ptr(),
                  asel.data_ptr(), scale(2).data_ptr(), shift(2).data_ptr(), mean(2).data_ptr(), rstd(2).data_ptr(),
                  S[2, 0].data_ptr(), S[2, 1].data_ptr(), dH2.data_ptr(), dW3.data_ptr(), S[1, 0].data_ptr(),
                  S[1, 1].data_ptr(), ws_f.data_ptr(), ws_d.data_ptr())
        else:
            _call("butd_sa_dz_last", X, B, np_, ns, C3, Z3.data_ptr(), d_out.data_ptr(), zsel.data_ptr(),
                  asel.data_ptr(), g3.data_ptr(), scale(2).data_ptr(), shift(2).data_ptr(), mean(2).data_ptr(),
                  rstd(2).data_ptr(), S[2, 0].data_ptr(), S[2, 1].data_ptr(), tr)
            dZ3 = Z3                                        # overwritten in place
            _gemm([_wgrad(dZ3, Z2, dW3, None, P, C3, C2, b_affine=(scale(1), shift(1))),
                   _dgrad(dZ3, w3, dH2, P, C3, C2)], X)
            # ---- layer 2
            _call("butd_sa_mask_stats", X, P, C2, dH2.data_ptr(), Z2.data_ptr(), scale(1).data_ptr(),
                  shift(1).data_ptr(), mean(1).data_ptr(), rstd(1).data_ptr(), S[1, 0].data_ptr(),
                  S[1, 1].data_ptr())
        first_lin = lin and not need_dfeat and Kp == 8 and _FIRST_LIN[0]
        d_feats = None
        assert not ctx.no_z1 or (first_lin and C1 == 64 and C2 == 64), "Z1 was not written: its backward needs butd_sa_mid_first_bwd"
        if first_lin and (_MID_FIRST[0] or ctx.no_z1) and C1 == 64 and C2 == 64:
            # layers 2 and 1 in one pass over (g2, Z2, Z1, X): dW2, dW1 and layer 1's sums, nothing written per row
            nf, nd = _mid_scratch(P, C1, Kp)
            ws_f = torch.empty(nf, device=dev)
            ws_d = torch.empty(nd, dtype=torch.float64, device=dev)
            _call("butd_sa_mid_first_bwd", X, P, C1, Kp, dH2.data_ptr(), Z2.data_ptr(), None if ctx.no_z1 else Z1.data_ptr(), X.data_ptr(),
                  g2.data_ptr(), scale(1).data_ptr(), shift(1).data_ptr(), mean(1).data_ptr(), rstd(1).data_ptr(),
                  S[1, 0].data_ptr(), S[1, 1].data_ptr(), scale(0).data_ptr(), shift(0).data_ptr(), mean(0).data_ptr(),
                  rstd(0).data_ptr(), w2.data_ptr(), w1.data_ptr(), dW2.data_ptr(), dW1.data_ptr(), S[0, 0].data_ptr(),
                  S[0, 1].data_ptr(), ws_f.data_ptr(), ws_d.data_ptr())
            return _SAMlpPool._finish(ctx, S, dW1, dW2, dW3, None, (C1, C2, C3), (s1, s2, s3), Cin)
        dH1 = torch.empty((P, C1), device=dev)
        mid_wide = lin and training and not first_lin and C1 == 128 and C2 == 128 and _MID_WIDE[0]
        if mid_wide:
            # layer 2 in one pass over (g2, Z2, Z1): dW2, layer 1's GATED gradient and its sums; no dZ2, no ungated dH1
            nf, nd = _wide_scratch(P, C2)
            ws_f = torch.empty(nf, device=dev)
            ws_d = torch.empty(nd, dtype=torch.float64, device=dev)
            _call("butd_sa_mid_wide_bwd", X, P, C2, dH2.data_ptr(), Z2.data_ptr(), Z1.data_ptr(), g2.data_ptr(),
                  scale(1).data_ptr(), shift(1).data_ptr(), mean(1).data_ptr(), rstd(1).data_ptr(), S[1, 0].data_ptr(),
                  S[1, 1].data_ptr(), scale(0).data_ptr(), shift(0).data_ptr(), mean(0).data_ptr(), rstd(0).data_ptr(),
                  w2.data_ptr(), dH1.data_ptr(), dW2.data_ptr(), S[0, 0].data_ptr(), S[0, 1].data_ptr(), ws_f.data_ptr(),
                  ws_d.data_ptr())
        else:
            _call("butd_sa_dz_mid", X, P, C2, dH2.data_ptr(), Z2.data_ptr(), g2.data_ptr(), scale(1).data_ptr(),
                  shift(1).data_ptr(), mean(1).data_ptr(), rstd(1).data_ptr(), S[1, 0].data_ptr(), S[1, 1].data_ptr(), tr)
        dZ2 = dH2
        # butd_sa_mask_stats of layer 1 as an epilogue mode of the product that writes dH1 (butd_gemm_problem.c_bn_*); the
        # 10^3..10^4 row tiles of a column add into 16 private copies of the sums (col_slots), folded below
        fuse1 = _FUSE_STATS[0] and not first_lin and C1 % 4 == 0 and not mid_wide
        if fuse1:
            Sx = zeros((16, 2, C1), dtype=torch.float64, device=dev)
            bn1 = dict(c_bn=(Z1, aff[0, 0], aff.stride(1), 0.0, 0), col_stats=(Sx[0, 0], Sx[0, 1]), col_slots=(16, 2 * C1))
        else:
            bn1 = {}
        if not mid_wide:
            _gemm([_wgrad(dZ2, Z1, dW2, None, P, C2, C1, b_affine=(scale(0), shift(0))),
                   _dgrad(dZ2, w2, dH1, P, C2, C1, **bn1)], X)
        # ---- layer 1
        if first_lin:
            # no input gradient wanted (SA1): the sums and dW1 from one pass over (dH1, Z1, X) -- no dZ1, no thin product
            nf, nd = _first_scratch(P, C1, Kp)
            ws_f = torch.empty(nf, device=dev)
            ws_d = torch.empty(nd, dtype=torch.float64, device=dev)
            _call("butd_sa_first_bwd", X, P, C1, Kp, dH1.data_ptr(), Z1.data_ptr(), X.data_ptr(), scale(0).data_ptr(),
                  shift(0).data_ptr(), mean(0).data_ptr(), rstd(0).data_ptr(), w1.data_ptr(), dW1.data_ptr(),
                  S[0, 0].data_ptr(), S[0, 1].data_ptr(), ws_f.data_ptr(), ws_d.data_ptr())
        else:
            if fuse1:
                S[0, :, :C1] = Sx.sum(0)
            elif not mid_wide:
                _call("butd_sa_mask_stats", X, P, C1, dH1.data_ptr(), Z1.data_ptr(), scale(0).data_ptr(),
                      shift(0).data_ptr(), mean(0).data_ptr(), rstd(0).data_ptr(), S[0, 0].data_ptr(),
                      S[0, 1].data_ptr())
            if ctx.lin1 is not None:
                # by linearity: one pass over (g1, Z1) along the inverted lists -> T (B N x C1) and dWx; the products over
                # the level's N points finish the layer (no dZ1, no grouped input gradient)
                xyz, new_xyz, feats_pm, Wf, radius, normalize, _ = ctx.lin1
                start, lst = ctx.inv if ctx.inv is not None else inverse_index(idx, N)
                import ctypes
                nws = ctypes.c_long(0)
                _hiplib.check(_lib.butd_sa_first_linear_bwd_scratch(B, N, C1, ctypes.byref(nws)), "butd_sa_first_linear_bwd_scratch")
                ws_l = torch.empty(nws.value, device=dev)
                T = torch.empty((B * N, C1), device=dev)
                dWx = torch.empty((C1, 3), device=dev)
                _call("butd_sa_first_linear_bwd", X, B, N, np_, ns, C1, xyz.data_ptr(), new_xyz.data_ptr(), start.data_ptr(),
                      lst.data_ptr(), radius, int(normalize), dH1.data_ptr(), Z1.data_ptr(), g1.data_ptr(),
                      scale(0).data_ptr(), shift(0).data_ptr(), mean(0).data_ptr(), rstd(0).data_ptr(), S[0, 0].data_ptr(),
                      S[0, 1].data_ptr(), tr, T.data_ptr(), dWx.data_ptr(), 3, ws_l.data_ptr())
                dWf = zeros((C1, C), device=dev)
                probs = [_wgrad(T, feats_pm, dWf, None, B * N, C1, C)]
                if need_dfeat:
                    d_feats = torch.empty((B, N, C), device=dev)
                    probs.append(_dgrad(T, Wf, d_feats, B * N, C1, C))
                _gemm(probs, X)
                if _pending_folds:
                    _flush_folds(S)
                dW1 = torch.cat((dWx, dWf), 1)
                return _SAMlpPool._finish(ctx, S, dW1, dW2, dW3, d_feats, (C1, C2, C3), (s1, s2, s3), Cin)
            _call("butd_sa_dz_mid", X, P, C1, dH1.data_ptr(), Z1.data_ptr(), g1.data_ptr(), scale(0).data_ptr(),
                  shift(0).data_ptr(), mean(0).data_ptr(), rstd(0).data_ptr(), S[0, 0].data_ptr(), S[0, 1].data_ptr(), tr)
            dZ1 = dH1
            if need_dfeat:
                dX = torch.empty((P, Kp), device=dev)
                _gemm([_wgrad(dZ1, X, dW1, None, P, C1, Kp), _dgrad(dZ1, w1, dX, P, C1, Kp)], X)
                if _GATHER[0] and C <= 256 and 256 % C == 0:
                    start, lst = ctx.inv if ctx.inv is not None else inverse_index(idx, N)
                    d_feats = torch.empty((B, N, C), device=dev)
                    _call("butd_sa_gather_rows", X, B, N, C, dX.data_ptr(), Kp, start.data_ptr(), lst.data_ptr(),
                          d_feats.data_ptr())
                else:
                    d_feats = zeros((B, N, C), device=dev)
                    _call("butd_sa_scatter_rows", X, B, N, np_, ns, C, dX.data_ptr(), Kp, idx.data_ptr(),
                          d_feats.data_ptr())
            else:
                _gemm([_wgrad(dZ1, X, dW1, None, P, C1, Kp)], X)
        return _SAMlpPool._finish(ctx, S, dW1, dW2, dW3, d_feats, (C1, C2, C3), (s1, s2, s3), Cin)

    @staticmethod
    def _finish(ctx, S, dW1, dW2, dW3, d_feats, widths, shapes, Cin):
        if _pending_folds:          # (dW1 is re-laid out below: the split-K slabs of the dense path have to be folded first)
            _flush_folds(S)
        C1, C2, C3 = widths
        s1, s2, s3 = shapes
        dW1 = dW1[:, :Cin]
        Sf = S.float()
        # S1 = sum g, S2 = sum g*zhat with zhat from the statistics the forward used (batch or running):
        # dbeta and dgamma in training AND eval mode
        dgs = [Sf[l, 1, :c].contiguous() for l, c in ((0, C1), (1, C2), (2, C3))]
        dbs = [Sf[l, 0, :c].contiguous() for l, c in ((0, C1), (1, C2), (2, C3))]
        return (None, d_feats, None, None, None, None, None, None,
                dW1.reshape(s1), dgs[0], dbs[0], None, None, None, None,
                dW2.view(s2), dgs[1], dbs[1], None, None, None, None,
                dW3.view(s3), dgs[2], dbs[2], None, None, None, None,
                None, None, None, None)


def supported(module, xyz, features_pm):
    """The fused pipeline covers the backbone's configuration: 3-layer BN MLP, max-pool, use_xyz."""
    from torch import nn
    mlp = module.mlp_module
    if not (xyz.is_cuda and module.pooling == "max" and module.use_xyz and module.npoint is not None
            and len(mlp) == 3 and not module.ret_unique_cnt):
        return False
    for layer in mlp:
        if not (hasattr(layer, "bn") and layer.conv.bias is None):
            return False
        c = layer.conv.out_channels
        if c > 256 or 256 % c:
            return False
    return 256 % module.nsample == 0 or module.nsample in (16, 32, 64)


def fused_eval_supported(module):
    """butd_sa_fused_eval: hidden widths <= 128, output <= 256, all multiples of 32, nsample 16 / 32 / 64."""
    cs = [l.conv.out_channels for l in module.mlp_module]
    return (module.nsample in (16, 32, 64) and all(c % 32 == 0 for c in cs) and cs[0] <= 128 and cs[1] <= 128
            and cs[2] <= 256 and _FUSED_EVAL[0])


def sa_fused_eval(module, xyz, new_xyz, idx, features_pm=None, feat_offset=0):
    """Inference: the whole level in ONE kernel (include/butd_sa.h: butd_sa_fused_eval) -- neighbourhood tiles in
    LDS, three BatchNorm-folded 1x1 convolutions LDS -> MFMA -> LDS, max-pool; nothing but the pooled
    (B, npoint, C3) features is written (SA1 at 8 x 50 000 points: 8 MB instead of 34 + 268 + 268 + 537 MB of
    grouped input and layer outputs).  No autograd graph: callers use it under ``torch.no_grad()`` / when nothing
    upstream requires a gradient."""
    import ctypes
    layers = list(module.mlp_module)
    B, N, _ = xyz.shape
    np_, ns = idx.shape[1], idx.shape[2]
    C = 0 if features_pm is None else features_pm.shape[-1] - feat_offset
    dev = xyz.device
    ws, scales, shifts = [], [], []
    for l in layers:
        bn = l.bn.bn
        w = l.conv.weight.detach().reshape(l.conv.out_channels, -1).contiguous()
        sc = (bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)).contiguous()
        ws.append(w)
        scales.append(sc)
        shifts.append((bn.bias.detach() - bn.running_mean * sc).contiguous())
    assert ws[0].shape[1] == 3 + C
    c_out = (ctypes.c_int * 3)(*[w.shape[0] for w in ws])
    ldw = (ctypes.c_long * 3)(*[w.shape[1] for w in ws])
    arr = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
    C3 = ws[2].shape[0]
    out_pm = torch.empty((B, np_, C3), device=dev)
    out_cm = torch.empty((B, C3, np_), device=dev)
    xyz, new_xyz, idx = xyz.contiguous(), new_xyz.contiguous(), idx.contiguous()
    fptr = None
    stride = 0
    if features_pm is not None:
        features_pm = features_pm.contiguous()
        fptr = features_pm.data_ptr() + 4 * feat_offset
        stride = features_pm.shape[-1]
    _call("butd_sa_fused_eval", xyz, B, N, np_, ns, C, xyz.data_ptr(), new_xyz.data_ptr(), fptr, stride,
          idx.data_ptr(), float(module.radius), int(bool(module.normalize_xyz)), c_out, arr(ws), ldw, arr(scales),
          arr(shifts), out_pm.data_ptr(), out_cm.data_ptr())
    return out_cm, out_pm


def sa_mlp_pool(module, xyz, new_xyz, idx, features_pm=None, feat_offset=0, inv=None):
    """-> (new_features (B,C,npoint), new_features_pm (B,npoint,C)).  ``features_pm``: point-major
    (B,N,offset+C) tensor whose last C columns are the per-point features (for SA1 the raw point cloud
    with offset 3)."""
    if (not module.training and fused_eval_supported(module) and not torch.is_grad_enabled()):
        return sa_fused_eval(module, xyz, new_xyz, idx, features_pm, feat_offset)
    layers = list(module.mlp_module)
    bns = [l.bn.bn for l in layers]
    training = module.training
    momentum = bns[0].momentum if bns[0].momentum is not None else 0.1
    args = []
    for l, bn in zip(layers, bns):
        args += [l.conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                 bn.num_batches_tracked, bn.eps]
    stride = 0 if features_pm is None else features_pm.shape[-1]
    return _SAMlpPool.apply(xyz.contiguous(), None if features_pm is None else features_pm.contiguous(),
                            new_xyz.contiguous(), idx, module.radius, module.normalize_xyz, training,
                            momentum, *args, feat_offset, stride, *(inv if inv is not None else (None, None)))
