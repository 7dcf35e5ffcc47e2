"""Row-wise helper operators on the gfx950 kernels of include/butd_rowwise.h (one launch each instead of the stock
chains of 3-13 elementwise / reduction launches): ``F.normalize`` of the contrastive-alignment projections
(bdetr.py:263-268,289-293,300-305) with its backward, and the inverse-distance weights of the feature-propagation
modules (pointnet2_modules.py:392-396)."""
import torch
import torch.nn.functional as F

from . import _hiplib


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


class _L2Normalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        lib = _hiplib.load()
        y = torch.empty_like(x)
        cols = x.shape[-1]
        with torch.cuda.device(x.device):
            _hiplib.check(lib.butd_l2_normalize_fwd(x.numel() // cols, cols, x.data_ptr(), eps, y.data_ptr(),
                                                    _stream(x)), "butd_l2_normalize_fwd")
        ctx.save_for_backward(x)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        lib = _hiplib.load()
        g = g.contiguous()
        dx = torch.empty_like(x)
        cols = x.shape[-1]
        with torch.cuda.device(x.device):
            _hiplib.check(lib.butd_l2_normalize_bwd(x.numel() // cols, cols, x.data_ptr(), g.data_ptr(), ctx.eps,
                                                    dx.data_ptr(), _stream(x)), "butd_l2_normalize_bwd")
        return dx, None


def l2_normalize(x, eps=1e-12):
    """``F.normalize(x, p=2, dim=-1)``; one kernel forward, one backward on the fused backend."""
    from . import attention_blocks
    if (attention_blocks.get_backend() == "hip" and x.is_cuda and x.dtype == torch.float32
            and x.shape[-1] % 4 == 0 and x.shape[-1] <= 1024):
        return _L2Normalize.apply(x.contiguous(), float(eps))
    return F.normalize(x, p=2, dim=-1, eps=eps)


def three_nn_weights(dist2):
    """(B, n, 3) squared 3-NN distances -> inverse-distance interpolation weights (B, n, 3), equal (to an ulp) to
    ``r / r.sum(-1, keepdim=True)`` with ``r = 1 / (sqrt(dist2) + 1e-8)``."""
    lib = _hiplib.load()
    dist2 = dist2.contiguous()
    w = torch.empty_like(dist2)
    with torch.cuda.device(dist2.device):
        _hiplib.check(lib.butd_three_nn_weights(dist2.numel() // 3, dist2.data_ptr(), None, w.data_ptr(),
                                                _stream(dist2)), "butd_three_nn_weights")
    return w
