"""-m gpu: BASELINE.json configs[4] (stress): 200 000-point scenes, nsample = 64 ball query, 512 queries,
128 text tokens, 2 scenes per GPU.  Index ops bit-exact against the oracle at that size; the whole model
forward + backward at that shape on the fused gfx950 path against the stock-torch maths of the same
modules (<= 2e-3 of the output scale), which also exercises the shape-dependent kernel configurations
(pruned FPS grid, GEMM tile choice, attention tails: 512 queries x 128 keys)."""
import copy
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from butd_detr_amd.synthetic_scenes import scannet_like_scene  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_index_ops_bit_exact_at_200k_points(oracle):
    from butd_detr_amd import pointnet2_ext as ext
    xyz = np.ascontiguousarray(scannet_like_scene(2024, 200000)[None, :, :3])
    ref = oracle.furthest_point_sampling(xyz, 2048, multithread=True)
    got = ext.furthest_point_sampling(dev(xyz), 2048).cpu().numpy()
    np.testing.assert_array_equal(got, ref)
    new_xyz = np.ascontiguousarray(np.take_along_axis(xyz, ref[..., None].astype(np.int64), 1))
    ref_idx = oracle.ball_query(new_xyz, xyz, 0.2, 64)
    got_idx = ext.ball_query(dev(new_xyz), dev(xyz), 0.2, 64).cpu().numpy()
    np.testing.assert_array_equal(got_idx, ref_idx)


def test_dense_cluster_scene_index_ops_bit_exact_vs_oracle(oracle):
    """The configs[4] scene WITH the 10x dense cluster (the case whose work depends on density: the pruned FPS's
    chunk re-evaluations and the grid ball query's cell occupancy) against the oracle -- the same clouds the
    train-mode test below feeds the model: SA1 (200 000 -> 2048, r = 0.2, 64 samples) of both scenes, and SA2
    (2048 -> 1024, r = 0.4, 32 samples) on the sampled centres.  sampling_gpu.cu:74-178, ball_query_gpu.cu:14-49."""
    from butd_detr_amd import pointnet2_ext as ext
    from butd_detr_amd.train_step import synthetic_batch
    inputs, _ = synthetic_batch(2, torch.device("cuda", 0), seed=311, n_points=200000, tokens=128, dense_cluster=True)
    xyz_d = inputs["point_clouds"][..., :3].contiguous()
    xyz = xyz_d.cpu().numpy()
    ref = oracle.furthest_point_sampling(xyz, 2048, multithread=True)
    got = ext.furthest_point_sampling(xyz_d, 2048).cpu().numpy()
    np.testing.assert_array_equal(got, ref)
    new_xyz = np.ascontiguousarray(np.take_along_axis(xyz, ref[..., None].astype(np.int64), 1))
    ref_idx = oracle.ball_query(new_xyz, xyz, 0.2, 64)
    got_idx = ext.ball_query(dev(new_xyz), xyz_d, 0.2, 64).cpu().numpy()
    np.testing.assert_array_equal(got_idx, ref_idx)
    ref2 = oracle.furthest_point_sampling(new_xyz, 1024)
    got2 = ext.furthest_point_sampling(dev(new_xyz), 1024).cpu().numpy()
    np.testing.assert_array_equal(got2, ref2)
    xyz2 = np.ascontiguousarray(np.take_along_axis(new_xyz, ref2[..., None].astype(np.int64), 1))
    np.testing.assert_array_equal(ext.ball_query(dev(xyz2), dev(new_xyz), 0.4, 32).cpu().numpy(),
                                  oracle.ball_query(xyz2, new_xyz, 0.4, 32))


def _close(a, b, tol, name, frac=1e-3):
    a, b = a.detach().float().cpu().numpy(), b.detach().float().cpu().numpy()
    scale = max(np.abs(b).max(), 1e-6)
    bad = np.abs(a - b) / scale > tol
    assert bad.mean() <= frac, f"{name}: {bad.sum()}/{bad.size} beyond {tol}; max {np.abs(a - b).max() / scale:.3e}"


def test_model_forward_backward_at_stress_shape():
    from butd_detr_amd import attention_blocks
    from butd_detr_amd.bdetr import BeaUTyDETR
    from butd_detr_amd.offline_text import offline_factory
    from butd_detr_amd.train_step import surrogate_loss, synthetic_batch
    try:
        torch.manual_seed(0)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = BeaUTyDETR(num_class=256, num_obj_class=485, input_feature_dim=3, num_queries=512,
                             num_decoder_layers=2, num_encoder_layers=1, self_position_embedding="loc_learned",
                             contrastive_align_loss=True, butd=True, self_attend=True,
                             text_encoder_factory=offline_factory(0)).cuda().eval()
        fused = copy.deepcopy(ref)
        inputs, targets = synthetic_batch(2, torch.device("cuda", 0), seed=77, n_points=200000, tokens=128)
        assert inputs["point_clouds"].shape == (2, 200000, 6)
        outs = {}
        for name, model, backend in (("torch", ref, "torch"), ("hip", fused, "hip")):
            attention_blocks.set_backend(backend)
            ep = model(inputs)
            loss = surrogate_loss(ep, targets)
            loss.backward()
            outs[name] = (ep, float(loss.detach()), {n: p.grad for n, p in model.named_parameters() if p.grad is not None})
        ep_t, loss_t, g_t = outs["torch"]
        ep_h, loss_h, g_h = outs["hip"]
        assert ep_h["last_sem_cls_scores"].shape == (2, 512, 256)
        assert ep_h["text_feats"].shape[1] == 128
        for key in ("fp2_features", "seed_features", "text_memory", "seeds_obj_cls_logits"):
            _close(ep_h[key], ep_t[key], 2e-3, key)
        # the 512 queries are the top-k seeds by objectness: two logits within rounding distance may
        # swap, which replaces whole query rows -- tolerate 1 % of them
        same = (ep_h["query_points_sample_inds"] == ep_t["query_points_sample_inds"]).float().mean().item()
        assert same >= 0.98, same
        for key in ("last_center", "last_pred_size", "last_sem_cls_scores", "proposal_proj_queries"):
            _close(ep_h[key], ep_t[key], 2e-3, key, frac=1e-2)
        assert abs(loss_h - loss_t) <= 2e-3 * max(abs(loss_t), 1.0)
        for n in ("backbone_net.sa1.mlp_module.layer0.conv.weight", "cross_encoder.layers.0.cross_layer.ffn_vl.0.weight",
                  "decoder.1.cross_v.in_proj_weight", "prediction_heads.1.center_residual_head.net.0.weight"):
            _close(g_h[n], g_t[n], 1e-2, n)
    finally:
        attention_blocks.set_backend("torch")


def test_stress_scene_with_dense_cluster_train_mode_full_depth():
    """configs[4] as SURVEY.md section 8(d) words it: 4x density (200 000 points in the same room) plus a 10x dense
    cluster, 512 queries, 128 tokens, 2 scenes -- the full 3 + 6 layer model in TRAIN mode (batch statistics,
    dropout 0), fused gfx950 path vs the stock-torch maths, strict mode (no silent fallback)."""
    from butd_detr_amd import attention_blocks
    from butd_detr_amd.bdetr import BeaUTyDETR
    from butd_detr_amd.offline_text import offline_factory
    from butd_detr_amd.train_step import surrogate_loss, synthetic_batch
    from tests.golden.cases import zero_dropout
    try:
        torch.manual_seed(0)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = BeaUTyDETR(num_class=256, num_obj_class=485, input_feature_dim=3, num_queries=512,
                             num_decoder_layers=6, num_encoder_layers=3, self_position_embedding="loc_learned",
                             contrastive_align_loss=True, butd=True, self_attend=True,
                             text_encoder_factory=offline_factory(0)).cuda()
        zero_dropout(ref.train())
        fused = copy.deepcopy(ref)
        inputs, targets = synthetic_batch(2, torch.device("cuda", 0), seed=311, n_points=200000, tokens=128,
                                          dense_cluster=True)
        pc = inputs["point_clouds"]
        assert pc.shape == (2, 200000, 6)
        # the cluster is there: the densest 0.5 m cell holds several times the points of the median occupied cell
        cell = (pc[0, :, :3] * 2).floor().long()
        key = (cell[:, 0] + 64) * 16384 + (cell[:, 1] + 64) * 128 + (cell[:, 2] + 64)
        counts = torch.unique(key, return_counts=True)[1].float()
        assert float(counts.max()) > 8 * float(counts.median()), (float(counts.max()), float(counts.median()))
        outs = {}
        for name, model, backend in (("torch", ref, "torch"), ("hip", fused, "hip")):
            attention_blocks.set_backend(backend)
            attention_blocks.set_strict(backend == "hip")
            ep = model(inputs)
            loss = surrogate_loss(ep, targets)
            loss.backward()
            outs[name] = (ep, float(loss.detach()), {n: p.grad for n, p in model.named_parameters() if p.grad is not None})
        ep_t, loss_t, g_t = outs["torch"]
        ep_h, loss_h, g_h = outs["hip"]
        assert ep_h["last_sem_cls_scores"].shape == (2, 512, 256) and ep_h["text_feats"].shape[1] == 128
        for key in ("sa1_inds", "sa2_inds", "fp2_inds"):
            assert torch.equal(ep_h[key], ep_t[key]), key
        # (both backends share the HIP index ops: the indices themselves are pinned against the ORACLE on this very
        #  scene by test_dense_cluster_scene_index_ops_bit_exact_vs_oracle above)
        for key in ("fp2_features", "seed_features", "text_memory", "seeds_obj_cls_logits"):
            _close(ep_h[key], ep_t[key], 2e-3, key)
        same = (ep_h["query_points_sample_inds"] == ep_t["query_points_sample_inds"]).float().mean().item()
        assert same >= 0.98, same
        # (a swapped query replaces its whole row in every per-query output: compare the rows both runs agree on)
        agree = ep_h["query_points_sample_inds"] == ep_t["query_points_sample_inds"]
        for key in ("last_center", "last_pred_size", "last_sem_cls_scores", "proposal_proj_queries"):
            _close(ep_h[key][agree], ep_t[key][agree], 2e-3, key, frac=2e-3)
        assert abs(loss_h - loss_t) <= 2e-3 * max(abs(loss_t), 1.0)
        assert set(g_h) == set(g_t)
        for n in ("backbone_net.sa1.mlp_module.layer0.conv.weight", "backbone_net.sa2.mlp_module.layer2.conv.weight",
                  "cross_encoder.layers.2.cross_layer.ffn_vl.0.weight", "decoder.5.cross_v.in_proj_weight",
                  "prediction_heads.5.center_residual_head.net.0.weight"):
            # (backbone weight gradients sum 2.6e5 - 2.6e7 grouped positions through train-mode BatchNorm in fp32
            #  with different summation orders: 1.2e-2 observed on 2 of SA1's 384 first-layer weights)
            _close(g_h[n], g_t[n], 2e-2 if n.startswith("backbone_net") else 1e-2, n, frac=1e-3)
    finally:
        attention_blocks.set_strict(False)
        attention_blocks.set_backend("torch")
