"""not gpu: fan_out / unstack (butd_detr_amd/fan_out.py) on the CPU: plain repetition resp. the torch fallback of the
gradient sum -- the model code calls them unconditionally (encoder_decoder_layers.py / bdetr.py call sites of the
reference: 356-404, 277-299)."""
import torch

from butd_detr_amd import fan_out as fo


def test_fan_out_is_plain_repetition_on_cpu():
    x = torch.randn(2, 5, 8, requires_grad=True)
    parts = fo.fan_out(x, 4)
    assert len(parts) == 4 and all(p is x for p in parts)
    assert fo.fan_out(None, 3) == (None,) * 3


def test_sum_fallback_and_unstack_on_cpu():
    g = [torch.randn(3, 4) for _ in range(5)]
    torch.testing.assert_close(fo._sum([g[0], None, g[1], g[2], None, g[3], g[4]]), sum(g[1:], g[0]))
    assert fo._sum([None, None]) is None and fo._sum([None, g[2]]) is g[2]
    x = torch.randn(4, 3, 2, requires_grad=True)
    y = x.detach().clone().requires_grad_(True)
    w = torch.randn(4, 3, 2)
    parts = fo.unstack(x * 1.0)
    torch.stack(parts).mul(w).sum().backward()
    ((y * 1.0) * w).sum().backward()
    torch.testing.assert_close(x.grad, y.grad)
    p = fo.unstack(x.detach())                       # no gradient needed: plain slices
    assert len(p) == 4 and torch.equal(p[2], x[2])
