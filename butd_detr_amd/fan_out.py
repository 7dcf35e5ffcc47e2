"""Gradient fan-in of tensors that feed several blocks.

The reference hands one tensor to many consumers: a decoder layer's ``query_pos`` to its four attention blocks
(encoder_decoder_layers.py:356-404), the encoder's visual / text / box outputs to all six decoder layers
(bdetr.py:277-299), the visual position embedding to every encoder layer.  Autograd sums the n incoming gradients
pairwise -- n - 1 launches that each re-read the running sum.  ``fan_out(t, n)`` returns n aliases of ``t`` whose
gradients arrive together and are summed by ONE kernel (``butd_sum_tensors``, include/butd_optim.h).
Values and gradients are those of using ``t`` n times (the sum is evaluated left to right, as autograd's).
"""
import ctypes
import os

import torch

from . import _hiplib, switches

_ENABLED = [switches.flag("fan_out", True)]


def set_enabled(flag):
    """A/B switch: off = plain reuse of the tensor, autograd's pairwise gradient sums (tests/test_gpu_free_running.py)."""
    prev, _ENABLED[0] = _ENABLED[0], bool(flag)
    return prev


def _sum(grads):
    live = [g for g in grads if g is not None]
    if not live:
        return None
    if len(live) == 1:
        return live[0]
    live = [g.contiguous() for g in live]
    first = live[0]
    if (not first.is_cuda or first.dtype != torch.float32 or len(live) > 8
            or any(g.shape != first.shape or g.dtype != first.dtype or (g.data_ptr() & 15) for g in live)):
        out = live[0] + live[1]
        for g in live[2:]:
            out = out + g
        return out
    out = torch.empty_like(first)
    ptrs = (ctypes.c_void_p * len(live))(*[g.data_ptr() for g in live])
    with torch.cuda.device(first.device):
        err = _hiplib.load().butd_sum_tensors(len(live), ptrs, first.numel(), out.data_ptr(),
                                              torch.cuda.current_stream(first.device).cuda_stream)
    _hiplib.check(err, "butd_sum_tensors")
    return out


class _FanOut(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, n):
        ctx.set_materialize_grads(False)             # an unused alias contributes None, not a tensor of zeros
        return tuple(t.view_as(t) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        return _sum(grads), None


def fan_out(t, n):
    """n aliases of ``t`` (n >= 1); plain repetition when there is nothing to gain (no gradient, CPU, n < 3, or
    BUTD_AB=fan_out=0 / set_enabled(False): A/B hook)."""
    if (t is None or n < 3 or not t.is_cuda or not t.requires_grad or not torch.is_grad_enabled()
            or not _ENABLED[0]):
        return (t,) * n
    return _FanOut.apply(t, n)


class _Unstack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        ctx.set_materialize_grads(False)
        ctx.shape, ctx.dtype, ctx.device = t.shape, t.dtype, t.device
        return tuple(t[i] for i in range(t.shape[0]))

    @staticmethod
    def backward(ctx, *grads):
        if all(g is not None for g in grads):
            return torch.stack(grads)                    # one pass instead of (zero-fill + copy + add) per slice
        out = torch.zeros(ctx.shape, dtype=ctx.dtype, device=ctx.device)
        for i, g in enumerate(grads):
            if g is not None:
                out[i].copy_(g)
        return out


def unstack(t):
    """The slices ``t[0], t[1], ...`` whose gradients are put back together by one ``stack`` (autograd's
    ``select_backward`` writes each slice's gradient into its own zero tensor of the full shape and adds them up)."""
    if not t.requires_grad or not torch.is_grad_enabled() or not _ENABLED[0]:
        return tuple(t[i] for i in range(t.shape[0]))
    return _Unstack.apply(t)
