"""-m gpu: fan_out (butd_detr_amd/fan_out.py, butd_sum_tensors): n aliases of a tensor whose gradients are summed in
one pass give the values and gradients of using the tensor n times (what the reference does with query_pos,
encoder_decoder_layers.py:356-404, and with the encoder outputs, bdetr.py:277-299)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,n", [((8, 256, 288), 4), ((8, 1024, 288), 6), ((3, 7, 5), 3), ((2, 132, 288), 8)])
def test_fan_out_matches_plain_reuse(shape, n):
    from butd_detr_amd.fan_out import fan_out
    torch.manual_seed(n)
    x = torch.randn(*shape, device="cuda", requires_grad=True)
    y = x.detach().clone().requires_grad_(True)
    w = [torch.randn(*shape, device="cuda") for _ in range(n)]
    parts = fan_out(x * 1.0, n)                      # non-leaf, like the model's tensors
    assert len(parts) == n and all(torch.equal(p, x) for p in parts)
    sum((p * wi).sum() for p, wi in zip(parts, w)).backward()
    sum(((y * 1.0) * wi).sum() for wi in w).backward()
    torch.testing.assert_close(x.grad, y.grad, rtol=1e-6, atol=1e-6)


def test_unused_aliases_and_fallbacks():
    from butd_detr_amd.fan_out import fan_out
    x = torch.randn(4, 64, 288, device="cuda", requires_grad=True)
    a, b, c, d = fan_out(x * 1.0, 4)
    (a.sum() * 2 + c.sum() * 3).backward()           # b, d never used: their gradients are None
    torch.testing.assert_close(x.grad, torch.full_like(x, 5.0))
    assert fan_out(None, 4) == (None,) * 4
    t = torch.randn(3, device="cuda")                # no gradient needed: plain repetition
    assert all(p is t for p in fan_out(t, 5))
    with torch.no_grad():
        assert all(p is x for p in fan_out(x, 3))


def test_sum_kernel_rejects_bad_arguments():
    import ctypes
    from butd_detr_amd import _hiplib
    lib = _hiplib.load()
    x = torch.zeros(16, device="cuda")
    ptrs = (ctypes.c_void_p * 1)(x.data_ptr())
    s = torch.cuda.current_stream().cuda_stream
    assert lib.butd_sum_tensors(0, ptrs, 16, x.data_ptr(), s) != 0
    assert lib.butd_sum_tensors(9, ptrs, 16, x.data_ptr(), s) != 0
    assert lib.butd_sum_tensors(1, (ctypes.c_void_p * 1)(x.data_ptr() + 4), 8, x.data_ptr(), s) != 0   # unaligned
    assert lib.butd_sum_tensors(1, ptrs, 0, x.data_ptr(), s) == 0


def test_unstack_matches_indexing():
    from butd_detr_amd.fan_out import unstack
    torch.manual_seed(0)
    x = torch.randn(7, 8, 256, 64, device="cuda", requires_grad=True)
    y = x.detach().clone().requires_grad_(True)
    w = torch.randn(7, 8, 256, 64, device="cuda")
    parts = unstack(x * 1.0)
    assert len(parts) == 7 and all(torch.equal(p, x[i]) for i, p in enumerate(parts))
    torch.stack([parts[i] for i in (1, 6, 0, 2, 3, 4, 5)]).mul(w).sum().backward()      # re-stacked in another order
    torch.stack([(y * 1.0)[i] for i in (1, 6, 0, 2, 3, 4, 5)]).mul(w).sum().backward()
    torch.testing.assert_close(x.grad, y.grad, rtol=0, atol=0)
    z = torch.randn(3, 5, device="cuda", requires_grad=True)
    p = unstack(z * 1.0)
    p[1].sum().backward()                                                              # unused slices: zero gradient
    assert torch.equal(z.grad, torch.tensor([[0.0] * 5, [1.0] * 5, [0.0] * 5], device="cuda"))
