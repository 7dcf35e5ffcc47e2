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
        assert torch.equal(pre._slot.inds_cur, want)
        # ... and so is the rest of the prefetched plan (centres, neighbour lists, interpolation indices / weights)
        plan = pre_model.backbone_net.plan(batches[-1][0]["point_clouds"])
        for got, exp in zip(GraphedTrainStep._plan_pieces(pre._slot.inputs["backbone_plan"]),
                            GraphedTrainStep._plan_pieces(plan)):
            assert torch.equal(got, exp)
    finally:
        attention_blocks.set_backend("torch")


def test_backbone_with_a_precomputed_plan_equals_the_inline_backbone():
    """Pointnet2Backbone.plan = the coordinate-only part of the backbone (FPS, centres, ball queries, 3-NN weights);
    forward(plan=...) must produce what forward() produces: same indices, same centres, same features."""
    from butd_detr_amd import attention_blocks
    from butd_detr_amd.backbone_module import Pointnet2Backbone
    from butd_detr_amd.synthetic_scenes import uniform_cloud
    try:
        attention_blocks.set_backend("hip")
        torch.manual_seed(3)
        net = Pointnet2Backbone(input_feature_dim=3).cuda().train()
        pc = torch.from_numpy(uniform_cloud(seed=9, n_points=8192, batch=2)).cuda()
        a = net(pc, end_points={})
        b = net(pc, end_points={}, plan=net.plan(pc))
        for k in ("sa1_inds", "sa2_inds", "fp2_inds", "sa1_xyz", "sa2_xyz", "sa3_xyz", "sa4_xyz"):
            assert torch.equal(a[k], b[k]), k
        for k in ("sa1_features", "sa4_features", "fp2_features"):       # (BatchNorm sums by atomics: rounding)
            assert float((a[k] - b[k]).abs().max()) <= 1e-5 * float(a[k].abs().max()), k
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


def test_warmup_does_not_train_signatures_are_cached_and_the_scheduler_is_followed():
    """Advisor findings of round 1: (1) the warm-up steps of a capture must not move parameters, optimizer
    state or BatchNorm buffers; (2) a signature seen before must not be captured again; (3) a learning rate
    written into param_groups (the reference steps a scheduler every iteration, main_utils.py:438) must reach
    the captured update."""
    from butd_detr_amd import attention_blocks
    from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, synthetic_batch
    try:
        dev = torch.device("cuda", 0)
        a = synthetic_batch(2, dev, seed=5, n_points=4096, tokens=24)
        b = synthetic_batch(2, dev, seed=6, n_points=4096, tokens=16)        # another token length
        model = _model()
        opt = FlatAdamW(model, lr=1e-4, lr_backbone=1e-3)
        p0 = opt.flat_p.clone()
        bn0 = model.backbone_net.sa1.mlp_module.layer0.bn.bn.running_mean.clone()
        step = GraphedTrainStep(model, opt, warmup=3)
        captures = []
        orig = step._capture
        step._capture = lambda *args: (captures.append(1), orig(*args))[1]
        step(*a)
        assert float(opt.step_count) == 1.0                                   # 3 warm-up steps left no trace
        assert int(model.backbone_net.sa1.mlp_module.layer0.bn.bn.num_batches_tracked) == 1
        moved = (opt.flat_p - p0).abs().max()
        assert 0 < float(moved) <= 1.1e-3, float(moved)                       # ONE AdamW step: |dp| <= lr (+ decay)
        assert not torch.equal(model.backbone_net.sa1.mlp_module.layer0.bn.bn.running_mean, bn0)
        step(*b)
        step(*a)
        step(*b)
        assert len(captures) == 2 and float(opt.step_count) == 4.0
        # scheduler: lr -> 0 for every group: the next replay must leave the parameters where they are
        for g in opt.param_groups:
            g["lr"] = 0.0
        before = opt.flat_p.clone()
        step(*a)
        assert torch.equal(opt.flat_p, before)
        for g, lr in zip(opt.param_groups, (1e-4, 1e-3)):
            g["lr"] = lr
        step(*a)
        assert not torch.equal(opt.flat_p, before)
        # checkpoint round trip (main_utils.py:131-152)
        sd = opt.state_dict()
        m2 = _model()
        opt2 = FlatAdamW(m2, lr=5e-5, lr_backbone=5e-5)
        opt2.load_state_dict(sd)
        assert torch.equal(opt2.flat_m, opt.flat_m) and float(opt2.step_count) == float(opt.step_count)
        assert [g["lr"] for g in opt2.param_groups] == [g["lr"] for g in opt.param_groups]
    finally:
        attention_blocks.set_backend("torch")


def test_alternating_signatures_replay_their_own_gradients():
    """Advisor finding of round 2: every captured signature (slot) must replay with ITS gradient pointer table -- a
    table shared between slots handed slot A the gradient addresses of slot B's private pool.  Learning rate 0 and a
    pinned dropout counter: every visit of a batch reproduces the loss AND the packed gradients of its first visit,
    whichever slot replayed in between; the least recently used slot is dropped at ``max_slots``."""
    from butd_detr_amd import attention_blocks, fused_attention as fa
    from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, HungarianCriterion, synthetic_batch
    try:
        dev = torch.device("cuda", 0)
        batches = [synthetic_batch(2, dev, seed=5 + i, n_points=4096, tokens=t) for i, t in enumerate((24, 16, 20))]
        lens = [len(step_in[0]["text"]) for step_in in batches]
        model = _model()
        opt = FlatAdamW(model, lr=0.0, lr_backbone=0.0, weight_decay=0.0)
        step = GraphedTrainStep(model, opt, warmup=1, criterion=HungarianCriterion(num_decoder_layers=2), max_slots=2)
        ctr = fa.rng_counter(dev)
        seen = {}
        order = [0, 1, 0, 1, 1, 0, 2, 0, 2, 1, 0]          # (slot 1 is evicted when 2 arrives, re-captured later)
        for k, i in enumerate(order):
            ctr.fill_(100 + i)
            loss = float(step(*batches[i]))
            g = opt.flat_g.clone()
            assert len(step._slots) <= 2
            if i in seen:
                l0, g0 = seen[i]
                assert abs(loss - l0) <= 1e-5 * max(abs(l0), 1.0), (k, i, loss, l0)
                assert float((g - g0).abs().max() / g0.abs().max()) <= 1e-4, (k, i)
            else:
                seen[i] = (loss, g)
        sigs = {tuple(b[0]["point_clouds"].shape) for b in batches}
        assert len(seen) == 3 and len(sigs) == 1 and float(seen[0][1].abs().max()) > 0
        # the three batches really are three signatures (token lengths differ)
        assert len({s[1] for s in step._slots} | {None}) >= 2
    finally:
        attention_blocks.set_backend("torch")


def test_overlapped_exchange_equals_the_single_graph_step(tmp_path):
    """The two-piece capture (backward cut at the encoder outputs, the decoder-side bucket all-reduced while the
    encoder / backbone backward replays) must train exactly like the single-graph step.  World size 1 over
    gloo with the collectives forced (BUTD_FORCE_COLLECTIVE)."""
    import os
    import torch.distributed as dist
    from butd_detr_amd import attention_blocks
    from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, HungarianCriterion, synthetic_batch
    os.environ["BUTD_FORCE_COLLECTIVE"] = "1"
    dist.init_process_group("gloo", init_method=f"file://{tmp_path}/init", rank=0, world_size=1)
    try:
        dev = torch.device("cuda", 0)
        batches = [synthetic_batch(2, dev, seed=40 + i, n_points=4096, tokens=24) for i in range(3)]
        one, two = _model(), None
        two = copy.deepcopy(one)
        crit = lambda: HungarianCriterion(num_decoder_layers=2)
        # learning rate 0: identical parameters in every step, so the packed gradients must agree to the
        # rounding of the atomics (split-K weight gradients)
        zero = dict(lr=0.0, lr_backbone=0.0, weight_decay=0.0)
        s1 = GraphedTrainStep(one, FlatAdamW(one, **zero), warmup=1, criterion=crit(), overlap_exchange=False)
        s2 = GraphedTrainStep(two, FlatAdamW(two, **zero), warmup=1, criterion=crit(), overlap_exchange=True)
        assert s2.split and not s1.split and 0 < s2.optimizer.boundary_offset < s2.optimizer.flat_g.numel()
        for inp, tgt in batches:
            l1, l2 = float(s1(inp, tgt)), float(s2(inp, tgt))
            assert abs(l1 - l2) <= 1e-5 * max(abs(l1), 1.0), (l1, l2)
            g1, g2 = s1.optimizer.flat_g, s2.optimizer.flat_g
            assert float((g1 - g2).abs().max() / g1.abs().max()) <= 1e-4
            assert float(g1.abs().max()) > 0
        # and with the reference's learning rates the two keep training alike (Adam's m / sqrt(v) turns last-bit
        # gradient differences into a fraction of lr where |g| ~ 0: compare the losses and the bulk)
        for g in s1.optimizer.param_groups + s2.optimizer.param_groups:
            g["lr"], g["weight_decay"] = 1e-4, 5e-4
        for inp, tgt in batches:
            l1, l2 = float(s1(inp, tgt)), float(s2(inp, tgt))
            assert abs(l1 - l2) <= 1e-3 * max(abs(l1), 1.0), (l1, l2)
        d = (s1.optimizer.flat_p - s2.optimizer.flat_p).abs()
        # (every weight gradient is a split-K product whose slices add in arrival order: two replays of the SAME step
        #  differ in the last bit there, and Adam moves a parameter whose gradient is ~0 by up to lr either way; with
        #  the contrastive projections on the grouped GEMM too, 97.6 % of this small model's parameters agree to 1e-5)
        assert float((d < 1e-5).float().mean()) > 0.95 and float(d.max()) < 1e-3, (float(d.max()), float((d < 1e-5).float().mean()))
    finally:
        dist.destroy_process_group()
        os.environ.pop("BUTD_FORCE_COLLECTIVE", None)
        attention_blocks.set_backend("torch")
