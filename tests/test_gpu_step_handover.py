"""-m gpu: GraphedTrainStep._copy_many -- the tensors a step takes over from its caller in ONE launch (pointer table +
butd_gather_segments) -- copies exactly what ``dst.copy_(src)`` copies, for every dtype / layout a batch holds
(train_dist_mod.py:103-110: float clouds and boxes, bool masks, int64 ids and tokens), pair by pair falling back to
``copy_`` where the fast path does not apply."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_copy_many_equals_copy():
    from butd_detr_amd.train_step import GraphedTrainStep
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(0)
    mk = lambda *shape, dtype=torch.float32: (torch.randn(*shape, generator=g) * 100).to(dtype).to(dev)
    srcs = [mk(8, 50000, 6), mk(8, 132, 6), mk(8, 132) > 0, mk(8, 132, dtype=torch.int64), mk(8, 80, dtype=torch.int64),
            mk(8, 1, 3), mk(3, 5), mk(7, dtype=torch.int32), (mk(5) > 0),                    # 5 bools: not 4-byte granular
            mk(4, 6)[:, ::2],                                                               # not contiguous
            mk(16, 4097), torch.empty(0, device=dev)]
    dsts = [torch.full_like(s, 7) if s.dtype != torch.bool else torch.zeros_like(s) for s in srcs]
    dsts[9] = torch.zeros(4, 3, device=dev)
    step = GraphedTrainStep.__new__(GraphedTrainStep)          # (only the hand-over helper: no model, no capture)
    for _ in range(6):                                         # more calls than the ring of pinned tables has slots
        for d in dsts:
            d.zero_()
        step._copy_many(list(zip(dsts, srcs)))
        torch.cuda.synchronize()
        for d, s in zip(dsts, srcs):
            assert torch.equal(d, s)
        srcs = [s.clone() for s in srcs]                       # other source addresses in the next call
        srcs[0] += 1
