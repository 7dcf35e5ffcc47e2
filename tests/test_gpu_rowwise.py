"""-m gpu: the row-wise helper kernels (include/butd_rowwise.h) and the loss weighting kernel
(butd_loss_combine) against the stock op chains they replace (models/bdetr.py:263-268,
pointnet2/pointnet2_modules.py:392-396, models/losses.py:592-617)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture
def hip_backend():
    from butd_detr_amd import attention_blocks as ab
    prev = ab.get_backend()
    ab.set_backend("hip")
    yield
    ab.set_backend(prev)


@pytest.mark.parametrize("shape", [(7, 8, 256, 64), (8, 80, 64), (3, 5, 288), (1, 1, 4), (2, 1024)])
def test_l2_normalize_forward_backward(hip_backend, shape):
    from butd_detr_amd.rowwise import l2_normalize
    torch.manual_seed(sum(shape))
    x = torch.randn(*shape, device="cuda")
    x.view(-1, shape[-1])[0].zero_()                      # a zero row: y = 0, dx = g / eps
    x.view(-1, shape[-1])[-1].mul_(1e-14)                 # a row below eps
    x.requires_grad_(True)
    probe = torch.randn(*shape, device="cuda")
    y_ref = F.normalize(x, p=2, dim=-1)
    (g_ref,) = torch.autograd.grad((y_ref * probe).sum(), x)
    y = l2_normalize(x)
    (g,) = torch.autograd.grad((y * probe).sum(), x)
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)
    scale = float(g_ref.abs().max())
    np.testing.assert_allclose(g.cpu().numpy() / scale, g_ref.cpu().numpy() / scale, rtol=0, atol=2e-6)


def test_three_nn_weights_equal_the_op_chain():
    from butd_detr_amd.rowwise import three_nn_weights
    torch.manual_seed(1)
    d2 = torch.rand(8, 1024, 3, device="cuda") * 0.3
    d2[0, 0] = 0.0                                        # a coincident point: 1 / 1e-8
    dist = torch.sqrt(d2)
    r = 1.0 / (dist + 1e-8)
    want = r / torch.sum(r, dim=2, keepdim=True)
    got = three_nn_weights(d2)
    # same operations in the same order; the device's sqrt / divide may differ from ATen's kernels in the last ulp
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=4e-7, atol=0)


def test_fp_module_same_output_on_both_backends():
    from butd_detr_amd import attention_blocks as ab
    from butd_detr_amd.pointnet2_modules import PointnetFPModule
    torch.manual_seed(2)
    fp = PointnetFPModule(mlp=[256 + 128, 256, 256]).cuda().eval()
    unknown, known = torch.rand(2, 512, 3, device="cuda"), torch.rand(2, 128, 3, device="cuda")
    uf, kf = torch.randn(2, 128, 512, device="cuda"), torch.randn(2, 256, 128, device="cuda")
    outs = {}
    for backend in ("torch", "hip"):
        ab.set_backend(backend)
        try:
            with torch.no_grad():
                outs[backend] = fp(unknown, known, uf, kf)
        finally:
            ab.set_backend("torch")
    np.testing.assert_allclose(outs["hip"].cpu().numpy(), outs["torch"].cpu().numpy(), rtol=0, atol=2e-4)


def test_loss_combine_matches_the_expression():
    from butd_detr_amd.losses import _CombineLosses
    torch.manual_seed(3)
    terms = [torch.rand(7, device="cuda", requires_grad=True) for _ in range(4)]
    gen = torch.rand((), device="cuda", requires_grad=True)
    status = torch.zeros(7, 8, dtype=torch.int32, device="cuda")
    want = 8 * gen + (1.0 / 7) * (terms[0].sum() + 5 * terms[1].sum() + terms[2].sum() + terms[3].sum())
    want_g = torch.autograd.grad(want * 3.0, terms + [gen])
    loss, ce, bbox, giou, align = _CombineLosses.apply(*terms, gen, status, 8.0, 1.0 / 7, 5.0)
    got_g = torch.autograd.grad(loss * 3.0, terms + [gen])
    assert abs(float(loss) - float(want)) < 1e-5 * max(1.0, abs(float(want))), (float(loss), float(want))
    assert abs(float(bbox) - float(terms[1].sum())) < 1e-6 and abs(float(align) - float(terms[3].sum())) < 1e-6
    for a, b in zip(got_g, want_g):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-6)
    status[3, 2] = 1                                        # a failed assignment -> NaN loss (scipy would raise)
    loss, *_ = _CombineLosses.apply(*terms, gen, status, 8.0, 1.0 / 7, 5.0)
    assert torch.isnan(loss)
    loss, ce, *_ = _CombineLosses.apply(None, terms[1], None, None, None, None, 8.0, 0.5, 5.0)   # optional terms
    assert abs(float(loss) - 2.5 * float(terms[1].sum())) < 1e-5 and float(ce) == 0.0
