"""FlatAdamW checkpoints carry the parameter layout (main_utils.py:131-152 saves / restores optimizer state):
moments saved under one packing order are re-mapped by name, never applied to the wrong parameters."""
import pytest
import torch

from butd_detr_amd.train_step import FlatAdamW


class _Toy(torch.nn.Module):
    def __init__(self, pre=()):
        super().__init__()
        self.cross_encoder = torch.nn.Linear(4, 6)
        self.decoder = torch.nn.Linear(5, 3)
        self.backbone_net = torch.nn.Linear(3, 2)
        self.pre_boundary_prefixes = pre


def _fill(opt, seed):
    g = torch.Generator().manual_seed(seed)
    opt.flat_m.copy_(torch.rand(opt.flat_m.shape, generator=g))
    opt.flat_v.copy_(torch.rand(opt.flat_v.shape, generator=g))
    opt.step_count.fill_(7.0)


def _moments_by_name(opt):
    return {n: (opt.flat_m[o:o + m].clone(), opt.flat_v[o:o + m].clone()) for n, o, m in opt.layout}


def test_same_layout_round_trip():
    a, b = FlatAdamW(_Toy()), FlatAdamW(_Toy())
    _fill(a, 1)
    b.load_state_dict(a.state_dict())
    assert torch.equal(a.flat_m, b.flat_m) and torch.equal(a.flat_v, b.flat_v) and float(b.step_count) == 7.0


def test_other_ordering_is_remapped_by_name():
    a = FlatAdamW(_Toy())                                    # registration order
    b = FlatAdamW(_Toy(pre=("cross_encoder.",)))             # decoder first, cross_encoder last in group 0
    assert [e[0] for e in a.layout] != [e[0] for e in b.layout]
    _fill(a, 2)
    b.load_state_dict(a.state_dict())
    want, got = _moments_by_name(a), _moments_by_name(b)
    for n in want:
        assert torch.equal(want[n][0], got[n][0]) and torch.equal(want[n][1], got[n][1]), n


def test_rejects_checkpoints_it_cannot_match():
    a, b = FlatAdamW(_Toy()), FlatAdamW(_Toy())
    sd = a.state_dict()
    del sd["layout"]
    with pytest.raises(ValueError, match="layout"):
        b.load_state_dict(sd)
    sd = a.state_dict()
    sd["layout"][0][0] = "renamed.weight"
    with pytest.raises(ValueError, match="different model"):
        b.load_state_dict(sd)
