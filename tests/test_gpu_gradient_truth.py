"""-m gpu: gradient precision of the fused path measured against a float64 run of the same modules.

Round 2 compared gradients of the train-mode 3+6-layer model with the CPU-fp32 reference golden and saw the fused
path at 1.6e-2 where stock torch on the GPU sat at 4e-3 -- "a 4x error nobody has located".  Against float64
(profiles/r03_gradient_error_vs_fp64.txt) the picture is: EVERY fp32 implementation of this model, stock torch
included, is 0.5-1.6e-2 (of a tensor's largest gradient) away from the truth on the last decoder layer; the fused path
is within ~2x of stock torch block by block; what is left are isolated discrete decisions -- a ReLU gate or a max-pool
winner of ONE channel deciding the other way for a pre-activation within rounding of zero -- which move that channel's
gradients by 1e-2..1e-1 (5 such channels among ~2 M gated activations; stock torch: 0-1).  Pinned here: the typical
(mean) error per parameter, the worst error per parameter outside a small budget of such flips.

Round 4: WHICH gates flip depends on the box (the forward sums BatchNorm statistics with float atomics, whose order
the hardware picks): on some boxes one gate of decoder layer 3's size head -- float64 pre-activation -1.7e-6 of a
column of scale 0.8, located by scratch/handoff_probe.py, profiles/r04_gradient_truth_flip.txt -- decides the other
way.  The size loss reaches only the few matched queries, so that ONE gate moves the MEAN error of the chain's first
BatchNorm weight to 3e-3: the mean criterion now tolerates such a tensor when it sits in at most four chains of the
model (a systematic loss of precision would show in many modules), under the same flip budget and a hard ceiling."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fusions(on):
    """The epilogue / ride-along fusions of rounds 4-6 on (the product) or off (round 3's launches: atomic LayerNorm
    backward, separate mask / statistics passes, dense set-abstraction backward with the grouped input, pairwise gradient
    sums, two-kernel attention backward, term-by-term criterion tail, per-block decoder key / value projections)."""
    from butd_detr_amd import fan_out, fused_attention as fa, fused_mlp, fused_sa, losses
    sa_flags = (fused_sa._LAST_LIN, fused_sa._LAST_FWD, fused_sa._FIRST_LIN, fused_sa._MID_FIRST, fused_sa._NO_Z1,
                fused_sa._FUSE_STATS, fused_sa._GATHER, fused_sa._MID_WIDE, fused_sa._FIRST_LINEAR, losses._TAIL)
    prev = (fa.set_ln_fold(on), fused_mlp.set_fuse_stats(on), fan_out.set_enabled(on), [f[0] for f in sa_flags],
            fa.set_long_keys(on), fa.set_decoder_kv_hoist(on))
    for f in sa_flags:
        f[0] = on
    return prev


def _restore_fusions(prev):
    from butd_detr_amd import fan_out, fused_attention as fa, fused_mlp, fused_sa, losses
    fa.set_ln_fold(prev[0]); fused_mlp.set_fuse_stats(prev[1]); fan_out.set_enabled(prev[2])
    for f, v in zip((fused_sa._LAST_LIN, fused_sa._LAST_FWD, fused_sa._FIRST_LIN, fused_sa._MID_FIRST, fused_sa._NO_Z1,
                     fused_sa._FUSE_STATS, fused_sa._GATHER, fused_sa._MID_WIDE, fused_sa._FIRST_LINEAR, losses._TAIL), prev[3]):
        f[0] = v
    fa.set_long_keys(prev[4])
    fa.set_decoder_kv_hoist(*prev[5])


@pytest.mark.parametrize("fusions", [True, False], ids=["product path", "round-3 launches (every later fusion off)"])
def test_fused_gradients_against_float64_truth(fusions):
    """``fusions=False`` (round 4's review): the same budget holds with the LayerNorm fold, the in-epilogue mask / statistics
    passes, the linear set-abstraction backward and the one-launch gradient fan-in switched OFF -- the gate / arg-max
    flips are a property of the fp32 forward's rounding, not of those fusions (a lifetime bug in one of them would
    show up as a difference between the two runs' budgets)."""
    from butd_detr_amd import attention_blocks, pointnet2_ext, pointnet2_utils
    from tests import grad_truth
    prev = _fusions(fusions)
    try:
        grad_truth.FIXED.clear()
        truth, _ = grad_truth.run("cpu", torch.float64, "torch")
        torch32, _ = grad_truth.run("cuda", torch.float32, "torch")
        hip32, ep = grad_truth.run("cuda", torch.float32, "hip")
    finally:
        _restore_fusions(prev)
        attention_blocks.set_backend("torch")
        pointnet2_utils._ext = pointnet2_ext
    top = max(float(t.abs().max()) for t in truth.values())
    flips, shifted, checked = [], [], 0
    for n, t in truth.items():
        scale = float(t.abs().max())
        if scale < 1e-6 * top:            # a bias in front of a BatchNorm: the true gradient is exactly 0
            continue
        checked += 1
        eh = (hip32[n] - t).abs() / scale
        et = (torch32[n] - t).abs() / scale
        # typical error: within 3x of stock torch (+ a floor for tensors torch happens to get to the last bit;
        # observed worst: 2.5x on SA1's first BatchNorm weight, a sum over 262 144 grouped rows)
        if float(eh.mean()) > 3.0 * float(et.mean()) + 5e-4:
            # a flip under a sparse loss (few rows carry gradient): the whole tensor moves -- ceiling 2e-2, and see below
            shifted.append((n, float(eh.mean()), float(et.mean())))
            flips.append((n, float(eh.max()), float(et.max())))
        elif float(eh.max()) > max(3.0 * float(et.max()), 6e-3):
            flips.append((n, float(eh.max()), float(et.max())))
        assert float(eh.mean()) <= 2e-2, (n, float(eh.mean()), float(et.mean()))
        assert float(eh.max()) <= 0.25, (n, float(eh.max()))
    assert checked > 500
    # discrete gate / arg-max flips: each touches the parameters of one chain (weight, BatchNorm weight / bias of the
    # layers below it: 6 tensors of a head, up to 27 of the backbone when the decision sits in SA3).  WHICH decisions
    # flip is a lottery of the forward's rounding (round 4: the forward that no longer writes Z3 sums its products in
    # another order -- same backbone flips, one more head chain: 22 tensors on some boxes, 27 on others, 14 before;
    # scratch/grad_truth_flips.py, profiles/r04_gradient_truth_flip.txt), so the budget counts chains, and tensors
    # only loosely (~7 % of the parameters)
    flip_chains = {"backbone_net" if n.startswith("backbone_net") else (n.split(".net.")[0] if ".net." in n else n.rsplit(".", 2)[0])
                   for n, _, _ in flips}
    assert len(flip_chains) <= 8 and len(flips) <= 40, (sorted(flip_chains), flips)
    # tensors whose TYPICAL error left the 3x band: only as the trace of a flip, i.e. confined to one or two chains
    chains = {n.split(".net.")[0] if ".net." in n else n.rsplit(".", 2)[0] for n, _, _ in shifted}
    # (three head chains on some boxes since the forward stopped writing Z3: heads 2 and 3 size, head 5 class scores)
    assert len(chains) <= 4 and len(shifted) <= 12, shifted
