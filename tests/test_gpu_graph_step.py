"""-m gpu: the hipGraph training step with the furthest-point-sampling chain and the frozen language
model prefetched for the NEXT batch on forked streams must train exactly like the step that runs them
in line -- over several
DIFFERENT batches (so a stale or mis-paired prefetch shows), including an unannounced batch."""
import copy
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model():
    from butd_detr_amd import attention_blocks
    from butd_detr_amd.bdetr import BeaUTyDETR
    from butd_detr_amd.offline_text import offline_factory
    attention_blocks.set_backend("hip")
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = BeaUTyDETR(num_class=256, num_obj_class=485, input_feature_dim=3, num_queries=64,
                       num_decoder_layers=2, num_encoder_layers=1, self_position_embedding="loc_learned",
                       contrastive_align_loss=True, butd=True, self_attend=True,
                       text_encoder_factory=offline_factory(0))
    m = m.cuda().train()
    for mod in m.modules():                      # identical arithmetic in both runs: no dropout draws
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if hasattr(mod, "dropout") and isinstance(getattr(mod, "dropout"), float):
            mod.dropout = 0.0
    return m


def test_prefetched_sampling_trains_like_inline_sampling():
    from butd_detr_amd import attention_blocks
    from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, synthetic_batch
    try:
        dev = torch.device("cuda", 0)
        batches = [synthetic_batch(2, dev, seed=100 + 7 * i, n_points=4096, tokens=24) for i in range(4)]
        ref_model = _model()
        pre_model = copy.deepcopy(ref_model)
        # learning rate 0: the per-batch losses are functions of (fixed weights, batch, sampled indices)
        # only, so they must agree to rounding -- a stale or mis-paired prefetch changes them visibly
        ref = GraphedTrainStep(ref_model, FlatAdamW(ref_model, lr=0.0, lr_backbone=0.0), warmup=1,
                               prefetch_sampling=False, prefetch_text=False)
        pre = GraphedTrainStep(pre_model, FlatAdamW(pre_model, lr=0.0, lr_backbone=0.0), warmup=1,
                               prefetch_sampling=True)
        ref_losses, pre_losses = [], []
        for i, (inp, tgt) in enumerate(batches):
            ref_losses.append(float(ref(inp, tgt)))
            # batch 2 arrives unannounced (the step before did not name it): falls back to sampling it
            nxt = batches[i + 1][0] if i + 1 < len(batches) and i != 1 else None
            pre_losses.append(float(pre(inp, tgt, next_inputs=nxt)))
        assert len({round(l, 3) for l in ref_losses}) == len(ref_losses)      # the batches do differ
        for a, b in zip(pre_losses, ref_losses):
            assert abs(a - b) <= 1e-4 * max(abs(b), 1.0), (pre_losses, ref_losses)
        # and the sampled indices the prefetching step consumed for the last batch are that batch's
        want = torch.cat([i.reshape(-1) for i in pre_model.backbone_net.sample(batches[-1][0]["point_clouds"])])
        assert torch.equal(pre.s_inds_cur, want)
    finally:
        attention_blocks.set_backend("torch")


def test_hungarian_criterion_inside_the_graph_equals_the_eager_step():
    """The reference's criterion (losses.compute_hungarian_loss: device-side assignment, static shapes) is
    captured in the hipGraph; replays over DIFFERENT batches (different numbers of targets, so different
    assignments) must give the losses of the plain eager evaluation of the same model and batch."""
    from butd_detr_amd import attention_blocks
    from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, HungarianCriterion, synthetic_batch
    try:
        dev = torch.device("cuda", 0)
        batches = [synthetic_batch(2, dev, seed=300 + 11 * i, n_points=4096, tokens=24) for i in range(3)]
        model = _model()
        crit = HungarianCriterion(num_decoder_layers=2)
        eager = []
        model.eval()                        # no BatchNorm statistics update between the two evaluations
        for inp, tgt in batches:
            with torch.no_grad():
                eager.append(float(crit(model(inp), crit.prepare(tgt))))
        step = GraphedTrainStep(model, FlatAdamW(model, lr=0.0, lr_backbone=0.0), warmup=1, criterion=crit)
        graphed = [float(step(inp, tgt)) for inp, tgt in batches]
        assert len({round(l, 3) for l in eager}) == len(eager)
        for a, b in zip(graphed, eager):
            assert abs(a - b) <= 2e-4 * max(abs(b), 1.0), (graphed, eager)
        n_boxes = [int((t["box_label_mask"] > 0).sum()) for _, t in batches]
        assert len(set(n_boxes)) > 1, n_boxes
    finally:
        attention_blocks.set_backend("torch")


def test_training_on_a_fixed_batch_reduces_the_hungarian_loss():
    """End-to-end sanity of the gradients through the criterion and the fused blocks: 25 graph-replayed steps on
    one fixed batch must bring the reference criterion's loss down."""
    from butd_detr_amd import attention_blocks
    from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, HungarianCriterion, synthetic_batch
    try:
        dev = torch.device("cuda", 0)
        inp, tgt = synthetic_batch(2, dev, seed=77, n_points=4096, tokens=24)
        model = _model().train()
        step = GraphedTrainStep(model, FlatAdamW(model, lr=2e-4, lr_backbone=2e-3), warmup=1,
                                criterion=HungarianCriterion(num_decoder_layers=2))
        losses = [float(step(inp, tgt, next_inputs=inp)) for _ in range(25)]
        assert all(l == l for l in losses), losses
        assert min(losses[-5:]) < 0.85 * losses[0], losses
    finally:
        attention_blocks.set_backend("torch")
