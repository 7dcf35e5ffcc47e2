"""Conv(k=1) + BatchNorm + ReLU (+ Dropout) chains on the gfx950 kernels of include/butd_mlp.h.

``mlp_chains`` runs G parallel chains that share one input -- the three ``ThreeLayerMLP`` of a
``ClsAgnosticPredictHead`` (models/modules.py:89-180), or G = 1 for ``PointsObjClsModule`` (:19-49),
``PositionEmbeddingLearned`` (:52-67) and the ``SharedMLP`` of ``PointnetFPModule``
(pointnet2_modules.py:371-416) -- forward AND backward, in training (batch statistics, running-stat
update, dropout) or eval mode.  Activations are position-major ``(P = B*L, C)`` matrices; the hidden
activations of the G chains sit side by side in one ``(P, G*H_l)`` matrix per layer.  Every layer is ONE
grouped MFMA GEMM launch whose epilogue accumulates the BatchNorm column sums and whose operand staging
applies the previous layer's BatchNorm + ReLU + Dropout, plus one tiny bookkeeping launch: a 3-chain
predict head is 5 launches forward and 10 backward, where the stock chain is ~33 + ~60 (Conv1d via
MIOpen with NCHW transposes, BatchNorm, ReLU, Dropout kernels and their backward).

A chain ends either in a plain convolution (heads, position embeddings) or -- ``out is None`` -- in
BatchNorm + ReLU (SharedMLP), whose activation is then materialised by ``butd_mlp_bn_relu_apply``.
"""
import os

import torch

from . import _hiplib, switches
from ._hiplib import BnSegment
from .fused_attention import _gemm, _problem, _slabbed, _stream, fold_scope, rng_counter, _site, zeros

_lib = _hiplib.load()
_FOLD_BN = [True]     # A/B switch of the in-product BatchNorm bookkeeping


_FUSE_STATS = [switches.flag("mlp_fuse_stats", True)]
_FUSE_DROP = [True]      # False: only chains without a Dropout (round 4's A/B: 23.42 -> 23.31 ms with it)


def set_fuse_stats(flag):
    prev, _FUSE_STATS[0] = _FUSE_STATS[0], bool(flag)
    return prev


def _call(name, ref, *args):
    with torch.cuda.device(ref.device):
        err = getattr(_lib, name)(*args, _stream(ref))
    _hiplib.check(err, name)


_THIN_SPLIT = [32]     # (scratch A/B: slices of a thin weight gradient; 8 = the rule of the square products)


def _split(M, N=0, K=0):
    # thin products (a 3- / 6-wide side: the box heads' output layers, the 6 -> 288 position embedding) run on the
    # element-wise staged kernel, ~1.5 us per 32-row slab in a handful of workgroups: more, shorter slices
    if 0 < min(N, K) <= 8:
        return max(1, min(_THIN_SPLIT[0], M // 32))
    return min(32, max(M // 256, min(8, M // 64), 1))      # see fused_attention._wgrad


class ChainSpec:
    """Static description of G chains: hidden widths per layer, output widths (empty: the chains end in
    BatchNorm + ReLU), buffers, flags (not tensors that need gradients: those go through
    ``_MlpChains.apply`` positionally)."""

    def __init__(self, G, Hs, outs, bn_buffers, eps, momentum, p_drop, training, site0):
        self.G, self.Hs, self.nh, self.outs = G, list(Hs), len(Hs), outs
        self.tail = not outs                 # no final convolution
        self.bn_buffers = bn_buffers         # [chain][layer] -> (running_mean, running_var, nbt)
        self.eps, self.momentum, self.p_drop = eps, momentum, p_drop
        self.training, self.site0 = training, site0


def _unpack(spec, params):
    G, nh = spec.G, spec.nh
    per = 4 * nh + (0 if spec.tail else 2)
    hidden = [[params[i * per + 4 * l:i * per + 4 * l + 4] for l in range(nh)] for i in range(G)]
    outp = None if spec.tail else [params[i * per + 4 * nh:i * per + 4 * nh + 2] for i in range(G)]
    return hidden, outp


class _MlpChains(torch.autograd.Function):
    """params: per chain, per hidden layer (w, bias|None, gamma, beta), then (w_out, b_out|None) unless
    the chains end in BatchNorm + ReLU."""

    @staticmethod
    def forward(ctx, x, spec, *params):
        G, nh, Hs = spec.G, spec.nh, spec.Hs
        P, Cin = x.shape
        dev = x.device
        hidden, outp = _unpack(spec, params)
        p = spec.p_drop if spec.training else 0.0
        Z = [torch.empty((P, G * H), device=dev) for H in Hs]
        Hmax = max(Hs)
        stats = zeros((nh, 2, G * Hmax), dtype=torch.float64, device=dev)
        aff = torch.empty((nh, 4, G * Hmax), device=dev)       # per layer: mean, rstd, scale, shift
        sl = lambda l, i: slice(i * Hs[l], (i + 1) * Hs[l])

        # Training: the product that CONSUMES a hidden layer computes that layer's BatchNorm scale / shift from the column
        # sums itself (butd_gemm_problem.a_bn_*: its first workgroup also writes mean / rstd / scale / shift for the
        # backward pass and updates the running statistics), so no bookkeeping launch sits between two products
        # (28 launches per step in the bench configuration).  Needs the float4 staging path (K <= 320, aligned rows).
        fold = [spec.training and _FOLD_BN[0] and Hs[l] <= 320 and (G * Hs[l]) % 4 == 0 and not (spec.tail and l == nh - 1)
                for l in range(nh)]

        def operand(l, i):
            """Input of layer l (l == nh: the output layer) of chain i: tensor, row stride, prologue."""
            if l == 0:
                return x, Cin, None, (0.0, 0), None
            s = sl(l - 1, i)
            drop = (p, spec.site0 + (l - 1) * G + i)
            if fold[l - 1]:
                rm, rv, nbt = spec.bn_buffers[i][l - 1]
                bn = (stats[l - 1, 0, s], stats[l - 1, 1, s], hidden[i][l - 1][2], hidden[i][l - 1][3], rm, rv, nbt,
                      aff[l - 1, 0, s], G * Hmax, P, spec.eps, spec.momentum)
                return Z[l - 1][:, s], G * Hs[l - 1], None, drop, bn
            return Z[l - 1][:, s], G * Hs[l - 1], (aff[l - 1, 2, s], aff[l - 1, 3, s]), drop, None

        for l in range(nh):
            H, GH = Hs[l], G * Hs[l]
            probs = []
            for i in range(G):
                w, b = hidden[i][l][0], hidden[i][l][1]
                a, lda, a_aff, a_drop, a_bn = operand(l, i)
                K = Cin if l == 0 else Hs[l - 1]
                probs.append(_problem(a, w, Z[l][:, sl(l, i)], P, H, K, (lda, 1), (K, 1), GH, bias=b,
                                      a_affine=a_aff, a_drop=a_drop, a_bn=a_bn,
                                      col_stats=(stats[l, 0, sl(l, i)], stats[l, 1, sl(l, i)])))
            _gemm(probs, x)
            if fold[l]:
                continue                       # (the next product does this layer's bookkeeping)
            segs = (BnSegment * _hiplib.MLP_MAX_SEGMENTS)()
            for i in range(G):
                rm, rv, nbt = spec.bn_buffers[i][l]
                segs[i] = BnSegment(hidden[i][l][2].data_ptr(), hidden[i][l][3].data_ptr(), rm.data_ptr(),
                                    rv.data_ptr(), None if nbt is None else nbt.data_ptr())
            _call("butd_mlp_bn_finalize", x, G, H, P, stats[l, 0].data_ptr(), stats[l, 1].data_ptr(), segs,
                  float(spec.eps), float(spec.momentum), int(spec.training), aff[l, 0].data_ptr(),
                  aff[l, 1].data_ptr(), aff[l, 2].data_ptr(), aff[l, 3].data_ptr())
        if spec.tail:
            GH = G * Hs[-1]
            out = torch.empty((P, GH), device=dev)
            _call("butd_mlp_bn_relu_apply", x, P, GH, GH, Z[-1].data_ptr(), aff[nh - 1, 2].data_ptr(),
                  aff[nh - 1, 3].data_ptr(), out.data_ptr())
            outs = [out]
        else:
            outs, probs = [], []
            for i in range(G):
                w, b = outp[i]
                a, lda, a_aff, a_drop, a_bn = operand(nh, i)
                o = torch.empty((P, spec.outs[i]), device=dev)
                probs.append(_problem(a, w, o, P, spec.outs[i], Hs[-1], (lda, 1), (Hs[-1], 1), spec.outs[i],
                                      bias=b, a_affine=a_aff, a_drop=a_drop, a_bn=a_bn))
                outs.append(o)
            _gemm(probs, x)
        ctx.save_for_backward(x, aff, *Z, *params)
        ctx.spec, ctx.p = spec, p
        return tuple(outs)

    @staticmethod
    @fold_scope
    def backward(ctx, *d_outs):
        spec, p = ctx.spec, ctx.p
        G, nh, Hs = spec.G, spec.nh, spec.Hs
        saved = ctx.saved_tensors
        x, aff = saved[0], saved[1]
        Z = saved[2:2 + nh]
        hidden, outp = _unpack(spec, saved[2 + nh:])
        P, Cin = x.shape
        dev = x.device
        Hmax = max(Hs)
        sl = lambda l, i: slice(i * Hs[l], (i + 1) * Hs[l])
        # one zero-filled slab for every weight / bias gradient (accumulated with atomics)
        sizes = []
        for i in range(G):
            for l in range(nh):
                w, b = hidden[i][l][0], hidden[i][l][1]
                sizes += [w.numel(), 0 if b is None else b.numel()]
            if not spec.tail:
                sizes += [outp[i][0].numel(), 0 if outp[i][1] is None else outp[i][1].numel()]
        slab = zeros(sum(sizes), device=dev)
        views, o = [], 0
        for n in sizes:
            views.append(slab[o:o + n] if n else None)
            o += n
        stride = 2 * nh + (0 if spec.tail else 2)
        dW = [[(views[i * stride + 2 * l], views[i * stride + 2 * l + 1]) for l in range(nh)]
              for i in range(G)]
        S = zeros((nh, 2, G * Hmax), dtype=torch.float64, device=dev)
        Sf = torch.empty((nh, 2, G * Hmax), device=dev)         # fp32 copies, written by butd_mlp_dz

        def operand(l, i):
            if l == 0:
                return x, Cin, None, (0.0, 0)
            s = sl(l - 1, i)
            return (Z[l - 1][:, s], G * Hs[l - 1], (aff[l - 1, 2, s], aff[l - 1, 3, s]),
                    (p, spec.site0 + (l - 1) * G + i))

        def wgrad(dy, ldy, N, l, i, dw, db):
            """dw[N,K] += dy[P,N]^T @ input_of_layer_l[P,K]  (+ db = column sums of dy)."""
            a, lda, b_aff, b_drop = operand(l, i)
            K = Cin if l == 0 else Hs[l - 1]
            return _slabbed(_problem(dy, a, dw, N, K, P, (1, ldy), (1, lda), K, bias_grad=db,
                                     ones_col=db is not None, accumulate=True, split_k=_split(P, N, K),
                                     b_affine=b_aff, b_drop=b_drop), dw, db)

        # The product that CREATES the gradient arriving at layer l's BatchNorm + ReLU applies the ReLU gate and leaves the
        # two column sums of the BatchNorm backward behind (butd_gemm_problem.c_bn_*): butd_mlp_mask_stats then has
        # nothing left to do for that layer (a dropout behind the activation included: the same counter hash).  Not for the
        # gradient that arrives from outside (a BatchNorm + ReLU tail).  BUTD_AB=mlp_fuse_stats=0: round 3's launches.
        p_of = lambda l: 0.0 if (spec.tail and l == nh - 1) else float(p)
        fused_stats = lambda l: _FUSE_STATS[0] and (_FUSE_DROP[0] or p_of(l) == 0.0)

        def bn_epilogue(l, i):
            if not fused_stats(l):
                return {}
            s_ = sl(l, i)
            return dict(c_bn=(Z[l][:, s_], aff[l, 0, s_], aff.stride(1), p_of(l), spec.site0 + l * G + i),
                        col_stats=(S[l, 0, s_], S[l, 1, s_]))

        stats_done = False
        GHl = G * Hs[-1]
        if spec.tail:      # d(out) is d(relu(bn(z))) of the last layer: it IS that layer's dH
            dH = d_outs[0].contiguous()
            if dH.data_ptr() == d_outs[0].data_ptr():
                dH = dH.clone()                       # mask_stats works in place
        else:
            # a gradient that arrives as a column slice of a wider matrix (the criterion hands the box heads the two
            # halves of d(pred_boxes)) is read in place through its row stride instead of being copied
            def rows(i, d):
                if d is None:
                    return zeros((P, spec.outs[i]), device=dev)
                if d.dim() == 2 and d.stride(1) == 1 and d.stride(0) >= d.shape[1] and d.dtype == torch.float32:
                    return d
                return d.contiguous()
            d_outs = [rows(i, d) for i, d in enumerate(d_outs)]
            dWo = [(views[i * stride + 2 * nh], views[i * stride + 2 * nh + 1]) for i in range(G)]
            dH = torch.empty((P, GHl), device=dev)
            probs = []
            for i in range(G):
                n_out, ldo = spec.outs[i], d_outs[i].stride(0)
                probs.append(wgrad(d_outs[i], ldo, n_out, nh, i, dWo[i][0], dWo[i][1]))
                probs.append(_problem(d_outs[i], outp[i][0], dH[:, sl(nh - 1, i)], P, Hs[-1], n_out,
                                      (ldo, 1), (1, Hs[-1]), GHl, **bn_epilogue(nh - 1, i)))
            _gemm(probs, x)
            stats_done = fused_stats(nh - 1)
        need_dx = ctx.needs_input_grad[0]
        dx = None
        for l in reversed(range(nh)):
            H, GH = Hs[l], G * Hs[l]
            # the activation of a BatchNorm+ReLU tail is not followed by a dropout
            p_l = p_of(l)
            if not stats_done:
                _call("butd_mlp_mask_stats", x, P, GH, GH, dH.data_ptr(), Z[l].data_ptr(), aff[l, 2].data_ptr(),
                      aff[l, 3].data_ptr(), aff[l, 0].data_ptr(), aff[l, 1].data_ptr(), p_l,
                      spec.site0 + l * G, H, rng_counter(dev).data_ptr(), S[l, 0].data_ptr(), S[l, 1].data_ptr())
            _call("butd_mlp_dz", x, P, GH, GH, dH.data_ptr(), Z[l].data_ptr(), aff[l, 2].data_ptr(),
                  aff[l, 0].data_ptr(), aff[l, 1].data_ptr(), S[l, 0].data_ptr(), S[l, 1].data_ptr(),
                  int(spec.training), Sf[l, 0].data_ptr(), Sf[l, 1].data_ptr())
            dZ = dH
            if l > 0:
                dH = torch.empty((P, G * Hs[l - 1]), device=dev)
            elif need_dx:
                dx = zeros((P, Cin), device=dev) if G > 1 else torch.empty((P, Cin), device=dev)
            probs = []
            for i in range(G):
                w = hidden[i][l][0]
                probs.append(wgrad(dZ[:, sl(l, i)], GH, H, l, i, dW[i][l][0], dW[i][l][1]))
                if l > 0:
                    Hp = Hs[l - 1]
                    probs.append(_problem(dZ[:, sl(l, i)], w, dH[:, sl(l - 1, i)], P, Hp, H, (GH, 1), (1, Hp),
                                          G * Hp, **bn_epilogue(l - 1, i)))
                elif need_dx:
                    probs.append(_problem(dZ[:, sl(l, i)], w, dx, P, Cin, H, (GH, 1), (1, Cin), Cin,
                                          accumulate=G > 1))
            _gemm(probs, x)
            stats_done = l > 0 and fused_stats(l - 1)
        grads = []
        for i in range(G):
            for l in range(nh):
                w, b = hidden[i][l][0], hidden[i][l][1]
                grads += [dW[i][l][0].view(w.shape), None if b is None else dW[i][l][1],
                          Sf[l, 1, sl(l, i)], Sf[l, 0, sl(l, i)]]
            if not spec.tail:
                grads += [dWo[i][0].view(outp[i][0].shape), dWo[i][1]]
        return (dx, None, *grads)


def mlp_chains(x_pm, chains, training):
    """x_pm (P, Cin) fp32 contiguous; ``chains``: list of (hidden, out, p_drop) with
    hidden = [(conv, bn), ...] (1x1 Conv1d/Conv2d and BatchNorm1d/2d modules), out = the final 1x1 conv
    or None (the chain ends in BatchNorm + ReLU).  All chains share depth and hidden widths.
    -> list of (P, out_channels) tensors (one (P, H_last) tensor when ``out`` is None)."""
    G = len(chains)
    nh = len(chains[0][0])
    Hs = [conv.out_channels for conv, _ in chains[0][0]]
    p_drop = chains[0][2]
    bn0 = chains[0][0][0][1]
    tail = chains[0][1] is None
    params, buffers, outs = [], [], []
    for hidden, out, p in chains:
        assert len(hidden) == nh and p == p_drop and (out is None) == tail
        bufs = []
        for l, (conv, bn) in enumerate(hidden):
            assert conv.out_channels == Hs[l] and all(k == 1 for k in conv.kernel_size) and bn.eps == bn0.eps
            assert bn.momentum == bn0.momentum and bn.affine and bn.track_running_stats
            params += [conv.weight, conv.bias, bn.weight, bn.bias]
            bufs.append((bn.running_mean, bn.running_var, bn.num_batches_tracked))
        buffers.append(bufs)
        if not tail:
            params += [out.weight, out.bias]
            outs.append(out.out_channels)
    assert G <= _hiplib.MLP_MAX_SEGMENTS and 2 * G <= 8 and all(H % 4 == 0 for H in Hs)
    assert not tail or G == 1
    site0 = _site[0] + 1
    _site[0] += G * nh
    momentum = 0.1 if bn0.momentum is None else bn0.momentum
    spec = ChainSpec(G, Hs, outs, buffers, bn0.eps, momentum, p_drop, bool(training), site0)
    return list(_MlpChains.apply(x_pm.contiguous(), spec, *params))
