"""Shared-MLP building blocks with the reference's parameter names.

Checkpoint compatibility is part of the drop-in contract (main_utils.py:122-141 loads
``strict=True``), so the module tree reproduces the key layout of pointnet2/pytorch_utils.py:
``<mlp>.layer{i}.conv.weight`` (1x1 Conv2d, **no bias when followed by BN**, kaiming-normal init),
``<mlp>.layer{i}.bn.bn.{weight,bias,running_mean,running_var,num_batches_tracked}`` (BN weight 1,
bias 0) and a parameter-free ``activation``.
"""
from torch import nn


class BatchNorm2d(nn.Sequential):
    """``bn.bn`` nesting of the reference (_BNBase, pytorch_utils.py:39-58)."""

    def __init__(self, channels):
        super().__init__()
        self.add_module("bn", nn.BatchNorm2d(channels))
        nn.init.constant_(self.bn.weight, 1.0)
        nn.init.constant_(self.bn.bias, 0.0)


class Conv2d(nn.Sequential):
    """1x1 conv [+ BN] [+ activation] (pytorch_utils.py:61-188, non-preact form)."""

    def __init__(self, in_size, out_size, *, bn=False, activation=True, bias=True):
        super().__init__()
        conv = nn.Conv2d(in_size, out_size, kernel_size=(1, 1), bias=bias and not bn)
        nn.init.kaiming_normal_(conv.weight)
        if conv.bias is not None:
            nn.init.constant_(conv.bias, 0.0)
        self.add_module("conv", conv)
        if bn:
            self.add_module("bn", BatchNorm2d(out_size))
        if activation:
            self.add_module("activation", nn.ReLU(inplace=True))


class SharedMLP(nn.Sequential):
    """Stack of ``layer{i}`` = Conv2d(1x1) + BN + ReLU over (B, C, npoint, nsample) tensors
    (pytorch_utils.py:11-36)."""

    def __init__(self, widths, *, bn=False):
        super().__init__()
        self.widths = list(widths)
        for i in range(len(widths) - 1):
            self.add_module(f"layer{i}", Conv2d(widths[i], widths[i + 1], bn=bn))
