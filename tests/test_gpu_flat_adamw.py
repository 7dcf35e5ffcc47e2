"""-m gpu: the packed AdamW kernel (+ in-kernel clip coefficient) vs torch.optim.AdamW + clip_grad_norm_."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_flat_adamw_matches_torch_adamw():
    from butd_detr_amd.train_step import FlatAdamW
    torch.manual_seed(0)

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone_net = torch.nn.Linear(37, 19)       # odd sizes: exercises the 4-float padding
            self.head = torch.nn.Sequential(torch.nn.Linear(19, 7), torch.nn.LayerNorm(7))

        def forward(self, x):
            return self.head(torch.relu(self.backbone_net(x)))

    ref = M().cuda()
    mine = copy.deepcopy(ref)
    named = list(ref.named_parameters())
    opt_ref = torch.optim.AdamW([
        {"params": [p for n, p in named if "backbone_net" not in n]},
        {"params": [p for n, p in named if "backbone_net" in n], "lr": 1e-2}], lr=1e-3, weight_decay=5e-4)
    opt = FlatAdamW(mine, lr=1e-3, lr_backbone=1e-2, weight_decay=5e-4)
    for step in range(5):
        x = torch.randn(32, 37, device="cuda")
        for m, o in ((ref, opt_ref), (mine, opt)):
            o.zero_grad(set_to_none=True)
            m(x).pow(2).sum().backward()
        torch.nn.utils.clip_grad_norm_([p for g in opt_ref.param_groups for p in g["params"]], 0.1)
        opt_ref.step()
        torch._foreach_copy_(opt.grad_views, [p.grad for p in opt.params])
        opt.clip_(0.1)
        opt.step()
        if step == 0:   # one step from identical state: the update rule itself, to fp32 rounding
            for (n, a), (_, b) in zip(ref.named_parameters(), mine.named_parameters()):
                np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=0,
                                           atol=1e-7, err_msg=n)
    # later steps: Adam's m/sqrt(v) is ill-conditioned where |g| ~ 0 (a 1e-8 parameter difference flips
    # such entries by a fraction of lr), so compare the bulk
    for (n, a), (_, b) in zip(ref.named_parameters(), mine.named_parameters()):
        d = (a - b).abs().detach().cpu().numpy()
        assert (d < 1e-5).mean() > 0.9 and d.max() < 1e-3, (n, d.max())


def test_gather_segments_equals_multi_tensor_copy():
    """FlatGradients.gather: one butd_gather_segments launch (include/butd_optim.h) vs torch._foreach_copy_, on
    odd sizes, unaligned slices and a re-used table."""
    from butd_detr_amd.train_step import FlatGradients
    torch.manual_seed(0)
    shapes = [(288, 288), (864,), (3,), (1,), (5000, 7), (64, 8), (4097,), (288,), (13, 17, 3)]
    params = [torch.nn.Parameter(torch.randn(*s, device="cuda")) for s in shapes]
    fg = FlatGradients(params)
    slab = torch.randn(sum(p.numel() for p in params) + 7, device="cuda")
    for rep in range(3):
        grads, off = [], 1 if rep == 1 else 0          # rep 1: sources at odd float offsets (unaligned)
        for p in params:
            grads.append(slab[off:off + p.numel()].view_as(p) * (rep + 1) if rep == 2 else
                         slab[off:off + p.numel()].view_as(p))
            off += p.numel()
        fg.flat.zero_()
        fg.gather(grads)
        want = torch.cat([g.reshape(-1) for g in grads])
        assert torch.equal(fg.flat, want), rep


def test_eager_use_collects_gradients_and_refuses_missing_ones():
    """Advisor finding of round 1: ``loss.backward(); opt.step()`` without a FlatGradients packing step used to
    update from an all-zero gradient buffer."""
    from butd_detr_amd.train_step import FlatAdamW
    torch.manual_seed(1)
    ref = torch.nn.Sequential(torch.nn.Linear(9, 5), torch.nn.Linear(5, 3)).cuda()
    mine = copy.deepcopy(ref)
    opt_ref = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=5e-4)
    opt = FlatAdamW(mine, lr=1e-3, weight_decay=5e-4)
    x = torch.randn(16, 9, device="cuda")
    for m, o in ((ref, opt_ref), (mine, opt)):
        o.zero_grad()
        m(x).pow(2).sum().backward()
        o.step()
    for a, b in zip(ref.parameters(), mine.parameters()):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=0, atol=1e-7)
    opt.zero_grad()
    mine[0](x).sum().backward()                 # the second layer gets no gradient
    with pytest.raises(RuntimeError, match="no gradient"):
        opt.step()
