"""-m gpu: the build's modules ON THE GPU (HIP operators + fused attention kernels) vs the golden vectors
captured from the reference (tests/golden/make_golden.py).  Tolerance: north_star's 1e-3 fp32 on
outputs, bit-exact indices."""
import os

import numpy as np
import pytest
import torch

from tests.golden import text_stub, weights
from tests.golden.cases import (backbone_inputs, bdetr_inputs, decoder_inputs, encoder_inputs,
                                probe)

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


# Bounds (of max|ref|), set in round 4 from what both backends actually reach on MI355X (gpurun_out/golden_errors.json,
# written by this file): encoder / decoder-layer goldens 0.9e-6 outputs, 1.7e-6 gradients; eval-mode backbone 6e-7 outputs,
# 3.4e-4 gradients; eval-mode model 2.9e-6; train-mode backbone 8e-6 outputs; train-mode 3 + 6-layer model 6.5e-5 (fused)
# / 1.3e-4 (stock ops) outputs.  Each bound is ~10x its observation (the rounds before had 1e-3 / 2e-3 everywhere).
OUT_TOL = 2e-5          # encoder, decoder layer, eval-mode backbone and model: outputs
GRAD_TOL = 2e-5         # encoder, decoder layer: gradients
BACKBONE_EVAL_GRAD_TOL = 1.5e-3
TRAIN_OUT_TOL = 1e-4    # train-mode backbone outputs (batch statistics reduced in another order than the CPU reference)
TRAIN6_OUT_TOL = 5e-4   # train-mode 3 + 6-layer model outputs
# train-mode model at the bench's size and batch (bdetr_50k_train.npz, 8 scenes), observed on MI355X
# (profiles/r06_golden_errors_bench_size_train.json): outputs 1.1e-5 (fused) / 1.5e-5 (stock ops) of scale; the 14 gradient tensors
# 1.1e-2 / 1.8e-2 max, worst mean 3.4e-3 / 3.5e-3 (with 2 scenes: 4.1e-5 / 8.7e-5, 2.6e-2 / 2.0e-2 max -- single ReLU-gate /
# max-pool flips: that tensor's mean was 2e-5 --, 2.0e-3 / 1.9e-3): the two backends are as far from the reference as from each
# other.  Bounds: 1.5-4x the observations.
BENCH_TRAIN_OUT_TOL, BENCH_TRAIN_GRAD_MAX, BENCH_TRAIN_GRAD_MEAN = 2e-4, 5e-2, 7e-3

_OBSERVED = []      # (test, max error, bound, share beyond the bound): written to gpurun_out/golden_errors.json


@pytest.fixture(scope="module", autouse=True)
def _dump_observed_errors():
    """What the bounds of this file are set FROM: every close() / grad_close() call's observed error next to its bound."""
    yield
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if _OBSERVED and os.path.isdir(os.path.join(root, "gpurun_out")):
        import json
        with open(os.path.join(root, "gpurun_out", "golden_errors.json"), "w") as f:
            json.dump(_OBSERVED, f, indent=0)


def _observe(err, tol):
    test = os.environ.get("PYTEST_CURRENT_TEST", "").split("::")[-1].split(" ")[0]
    _OBSERVED.append([test, float(err.max()), float(err.mean()), float(tol), float((err > tol).mean())])


def close(t, ref, tol=1e-3, outlier_frac=0.0, outlier_max=0.2):
    """|t - ref| <= tol * max|ref| everywhere, except for at most ``outlier_frac`` of the elements, which stay
    within ``outlier_max`` (max-pool arg-max flips between near-equal candidates reroute a few gradient entries)."""
    a = t.detach().float().cpu().numpy()
    scale = max(float(np.abs(ref).max()), 1e-6)
    _observe(np.abs(a - ref) / scale, tol)
    if outlier_frac:
        bad = np.abs(a - ref) / scale > tol
        assert bad.mean() <= outlier_frac, f"{bad.sum()} of {bad.size} elements beyond {tol}"
        assert np.abs(a - ref).max() / scale < outlier_max
        return
    np.testing.assert_allclose(a / scale, ref / scale, rtol=0, atol=tol)


def cuda(d):
    return {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in d.items()}


@pytest.fixture(params=["torch", "hip"])
def backend(request):
    from butd_detr_amd import attention_blocks
    prev = attention_blocks.get_backend()
    attention_blocks.set_backend(request.param)
    yield request.param
    attention_blocks.set_backend(prev)


def test_encoder_golden(backend):
    from butd_detr_amd.encoder_decoder_layers import BiEncoder, BiEncoderLayer
    g = load("encoder_small.npz")
    layer = BiEncoderLayer(288, dropout=0.1, activation="relu", n_heads=8, dim_feedforward=256,
                           self_attend_lang=True, self_attend_vis=True, use_butd_enc_attn=True)
    model = weights.fill_(BiEncoder(layer, 3), seed=11).cuda().eval()
    inp = cuda(encoder_inputs())
    for k in ("vis", "text", "pos", "boxes"):
        inp[k].requires_grad_(True)
    vis_out, text_out = model(inp["vis"], inp["pos"], inp["vis_mask"], inp["text"], inp["text_mask"],
                              {}, detected_feats=inp["boxes"], detected_mask=inp["box_mask"])
    close(vis_out, g["vis_out"], OUT_TOL)
    close(text_out, g["text_out"], OUT_TOL)
    ((vis_out * probe(vis_out.shape, 1).cuda()).sum() + (text_out * probe(text_out.shape, 2).cuda()).sum()).backward()
    close(inp["vis"].grad, g["g_vis"], GRAD_TOL)
    close(inp["text"].grad, g["g_text"], GRAD_TOL)
    close(inp["pos"].grad, g["g_pos"], GRAD_TOL)
    close(inp["boxes"].grad, g["g_boxes"], GRAD_TOL)
    p = dict(model.named_parameters())
    close(p["layers.0.cross_layer.cross_lv.in_proj_weight"].grad, g["g_l0_cross_lv_in_proj_weight"], GRAD_TOL)
    close(p["layers.2.self_attention_visual.self_attn.out_proj.weight"].grad,
          g["g_l2_self_attention_visual_out_proj_weight"], GRAD_TOL)
    close(p["layers.1.cross_layer.ffn_vl.0.weight"].grad, g["g_l1_ffn_vl_0_weight"], GRAD_TOL)
    close(p["layers.1.cross_layer.norm_d.weight"].grad, g["g_l1_norm_d_weight"], GRAD_TOL)


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_decoder_layer_golden(backend, mode):
    from butd_detr_amd.encoder_decoder_layers import BiDecoderLayer
    g = load(f"decoder_small_{mode}.npz")
    layer = BiDecoderLayer(288, n_heads=8, dim_feedforward=256, dropout=0.1 if mode == "eval" else 0.0,
                           activation="relu", self_position_embedding="loc_learned", butd=True)
    weights.fill_(layer, seed=12).cuda()
    layer.train(mode == "train")
    inp = cuda(decoder_inputs())
    for k in ("query", "vis", "text", "boxes"):
        inp[k].requires_grad_(True)
    out = layer(inp["query"], inp["vis"], inp["text"], inp["query_pos"], None, inp["text_mask"],
                detected_feats=inp["boxes"], detected_mask=inp["box_mask"])
    close(out, g["out"], OUT_TOL)
    (out * probe(out.shape, 3).cuda()).sum().backward()
    close(inp["query"].grad, g["g_query"], GRAD_TOL)
    close(inp["vis"].grad, g["g_vis"], GRAD_TOL)
    close(inp["text"].grad, g["g_text"], GRAD_TOL)
    close(inp["boxes"].grad, g["g_boxes"], GRAD_TOL)
    p = dict(layer.named_parameters())
    close(p["cross_v.in_proj_weight"].grad, g["g_cross_v_in_proj_weight"], GRAD_TOL)
    close(p["self_posembed.position_embedding_head.0.weight"].grad, g["g_self_posembed_0_weight"], GRAD_TOL)
    close(p["ffn.3.weight"].grad, g["g_ffn_3_weight"], GRAD_TOL)


def test_decoder_layer_golden_bench_shapes(backend):
    """The reference's decoder layer at the bench's per-layer shapes (6 scenes x 256 queries over 256 / 80 / 132 / 1024
    keys, train mode, dropout 0; tests/golden/make_golden.py golden_decoder_bench_shape): the shapes at which the
    attention backward runs as ONE pass (butd_attention_bwd_long_keys: 64-key chunks with dQ slabs at the three short
    sites, the 256-key plan at the 1024-seed site) -- pinned against the imported reference directly, not through the
    small goldens (whose 16 queries x 64 seeds take the two-kernel walk) or the full model's train6 golden."""
    from butd_detr_amd import _hiplib, fused_attention as fa
    from butd_detr_amd.encoder_decoder_layers import BiDecoderLayer
    from tests.golden.cases import decoder_bench_inputs
    g = load("decoder_256x1024_train.npz")
    layer = BiDecoderLayer(288, n_heads=8, dim_feedforward=256, dropout=0.0, activation="relu",
                           self_position_embedding="loc_learned", butd=True)
    weights.fill_(layer, seed=16).cuda()
    layer.train()
    inp = cuda(decoder_bench_inputs())
    if backend == "hip":        # the one-pass kernel serves all four sites of this batch
        lib = _hiplib.load()
        assert fa._long_keys[0]
        for lk in (256, 80, 132, 1024):
            assert int(lib.butd_attention_bwd_long_keys_scratch(6, 8, 256, lk, 36, 288)) >= 0
    for k in ("query", "vis", "text", "boxes"):
        inp[k].requires_grad_(True)
    out = layer(inp["query"], inp["vis"], inp["text"], inp["query_pos"], None, inp["text_mask"],
                detected_feats=inp["boxes"], detected_mask=inp["box_mask"])
    close(out[:, ::2], g["out_rows2"], OUT_TOL)
    (out * probe(out.shape, 5).cuda()).sum().backward()
    close(inp["query"].grad[:, ::2], g["g_query_rows2"], GRAD_TOL)
    close(inp["vis"].grad[:, ::8], g["g_vis_rows8"], GRAD_TOL)
    close(inp["vis"].grad.double().sum(1), g["g_vis_colsum"], GRAD_TOL)
    close(inp["text"].grad, g["g_text"], GRAD_TOL)
    close(inp["boxes"].grad[:, ::2], g["g_boxes_rows2"], GRAD_TOL)
    p = dict(layer.named_parameters())
    close(p["cross_v.in_proj_weight"].grad[::3], g["g_cross_v_in_proj_weight_rows3"], GRAD_TOL)
    close(p["self_attn.in_proj_weight"].grad[::3], g["g_self_attn_in_proj_weight_rows3"], GRAD_TOL)
    close(p["cross_d.out_proj.weight"].grad, g["g_cross_d_out_proj_weight"], GRAD_TOL)
    close(p["self_posembed.position_embedding_head.0.weight"].grad, g["g_self_posembed_0_weight"], GRAD_TOL)
    close(p["ffn.3.weight"].grad, g["g_ffn_3_weight"], GRAD_TOL)


@pytest.mark.parametrize("mode", ["eval", "train"])
@pytest.mark.parametrize("side", [False, True])
def test_decoder_layer_golden_hoisted_memory(mode, side):
    """The same reference vectors through the round-6 path of the full model: the layer's three memories projected by the
    hoisted node (fused_attention.decoder_memory, on the forked stream or not), its cross-attention blocks reading them,
    the memory-side input gradients and the K / V weight gradients from the hoisted node's backward."""
    from butd_detr_amd import attention_blocks, fused_attention as fa
    from butd_detr_amd.encoder_decoder_layers import BiDecoderLayer
    prev_backend = attention_blocks.get_backend()
    attention_blocks.set_backend("hip")
    prev = fa.set_decoder_kv_hoist(True, side=side)
    try:
        g = load(f"decoder_small_{mode}.npz")
        layer = BiDecoderLayer(288, n_heads=8, dim_feedforward=256, dropout=0.1 if mode == "eval" else 0.0,
                               activation="relu", self_position_embedding="loc_learned", butd=True)
        weights.fill_(layer, seed=12).cuda()
        layer.train(mode == "train")
        inp = cuda(decoder_inputs())
        for k in ("query", "vis", "text", "boxes"):
            inp[k].requires_grad_(True)
        shared = fa.decoder_memory([layer], [("text", "cross_l", inp["text"]), ("boxes", "cross_d", inp["boxes"]),
                                             ("seeds", "cross_v", inp["vis"])])
        assert shared is not None
        out = layer(inp["query"], inp["vis"].detach(), inp["text"].detach(), inp["query_pos"], None, inp["text_mask"],
                    detected_feats=inp["boxes"].detach(), detected_mask=inp["box_mask"], memory_kv=(shared, 0))
        shared.token = None
        close(out, g["out"], OUT_TOL)
        (out * probe(out.shape, 3).cuda()).sum().backward()
        close(inp["query"].grad, g["g_query"], GRAD_TOL)
        close(inp["vis"].grad, g["g_vis"], GRAD_TOL)
        close(inp["text"].grad, g["g_text"], GRAD_TOL)
        close(inp["boxes"].grad, g["g_boxes"], GRAD_TOL)
        p = dict(layer.named_parameters())
        close(p["cross_v.in_proj_weight"].grad, g["g_cross_v_in_proj_weight"], GRAD_TOL)
        close(p["self_posembed.position_embedding_head.0.weight"].grad, g["g_self_posembed_0_weight"], GRAD_TOL)
        close(p["ffn.3.weight"].grad, g["g_ffn_3_weight"], GRAD_TOL)
        assert all(q.grad is not None for q in layer.parameters())
    finally:
        fa.set_decoder_kv_hoist(*prev)
        attention_blocks.set_backend(prev_backend)


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_backbone_golden(mode):
    from butd_detr_amd.backbone_module import Pointnet2Backbone
    g = load(f"backbone_4096_{mode}.npz")
    net = weights.fill_(Pointnet2Backbone(input_feature_dim=3, width=1), seed=13).cuda()
    net.train(mode == "train")
    ep = net(backbone_inputs().cuda(), end_points={})
    for k in ("sa1_inds", "sa2_inds", "fp2_inds"):
        np.testing.assert_array_equal(ep[k].cpu().numpy(), g[k])
    np.testing.assert_array_equal(ep["sa1_xyz"].cpu().numpy(), g["sa1_xyz"])
    np.testing.assert_array_equal(ep["sa4_xyz"].cpu().numpy(), g["sa4_xyz"])
    otol = TRAIN_OUT_TOL if mode == "train" else OUT_TOL
    close(ep["sa1_features"][:, :, :64], g["sa1_features_head"], otol)
    close(ep["sa2_features"][:, :, :64], g["sa2_features_head"], otol)
    close(ep["sa4_features"], g["sa4_features"], otol)
    close(ep["fp2_features"][0], g["fp2_features_b0"], otol)
    (ep["fp2_features"] * probe(ep["fp2_features"].shape, 4).cuda()).sum().backward()
    p = dict(net.named_parameters())
    # weight gradients through up to 14 batch-norm layers: the GPU reduces the batch statistics in a
    # different order than the CPU reference run, so the deepest ones get a looser (1e-2) bound; observed in train mode:
    # max-pool winners that flip between near-equal candidates put a few elements beyond 1e-2 (profiles/r06_golden_errors.json:
    # none in three of the tensors, 0.006 % of sa3's at <= 1.3e-2, 0.065 % of fp2's at <= 4.4e-2; means 0.5-2.2e-3).  The count
    # moves from run to run with the arrival order of the float atomics (sa3's: 2 of 32 768 in four runs, 5 in a fifth): the
    # budgets are 0.05 % (16 elements of sa3's) resp. twice fp2's observed share, the outliers' bound ~2x the largest one seen
    gtol = 1e-2 if mode == "train" else BACKBONE_EVAL_GRAD_TOL
    frac, frac_fp2, omax = (5e-4, 1.3e-3, 0.1) if mode == "train" else (0.0, 0.0, 0.2)
    close(p["sa1.mlp_module.layer0.conv.weight"].grad, g["g_sa1_layer0_conv"], gtol, frac, omax)
    close(p["sa3.mlp_module.layer2.conv.weight"].grad, g["g_sa3_layer2_conv"], gtol, frac, omax)
    close(p["sa2.mlp_module.layer1.bn.bn.weight"].grad, g["g_sa2_layer1_bn_weight"], gtol, frac, omax)
    close(p["fp1.mlp.layer0.conv.weight"].grad, g["g_fp1_layer0_conv"], gtol, frac, omax)
    close(p["fp2.mlp.layer1.conv.weight"].grad, g["g_fp2_layer1_conv"], gtol, frac_fp2, omax)


def test_bdetr_golden(backend):
    import warnings
    from butd_detr_amd.bdetr import BeaUTyDETR
    g = load("bdetr_4096_eval.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = BeaUTyDETR(num_class=256, num_obj_class=485, input_feature_dim=3, num_queries=32,
                           num_decoder_layers=2, self_position_embedding="loc_learned",
                           contrastive_align_loss=True, butd=True, pointnet_ckpt=None, self_attend=True,
                           text_encoder_factory=text_stub.factory,
                           class_embeddings_path="/nonexistent/class_embeddings3d.npy")
    weights.fill_(model, seed=14, skip_prefixes=("text_encoder.",))
    model.cuda().eval()
    with torch.no_grad():
        ep = model(cuda(bdetr_inputs()))
    assert sorted(ep.keys()) == list(g["end_points_keys"])
    for k in ("seed_inds", "query_points_sample_inds", "text_attention_mask"):
        np.testing.assert_array_equal(ep[k].cpu().numpy(), g[k])
    for k in ("seeds_obj_cls_logits", "text_feats", "text_memory", "proj_tokens", "proposal_center",
              "proposal_pred_size", "proposal_proj_queries", "0head_center", "last_center",
              "last_pred_size", "last_sem_cls_scores", "last_proj_queries"):
        close(ep[k], g[k], OUT_TOL)
    close(ep["seed_features"][0], g["seed_features_b0"], OUT_TOL)


def test_bdetr_golden_at_the_bench_size(backend):
    """The REFERENCE model at the bench's size (2 scenes x 50 000 points, 256 queries, 3 + 6 layers, eval mode;
    tests/golden/make_golden.py golden_bdetr_bench_shape, index ops of the reference run = the CPU oracle): the pruned
    FPS and the grid ball query of level 1, the parallel FPS decision of levels 2-4, the fused set-abstraction levels at
    their full widths and every attention site at its real length against the reference directly -- not against this
    repo's own torch backend.  The query seeds are handed back (two objectness logits within rounding distance may swap
    a seed in or out of the top 256); without them at least 240 of the 256 seeds per scene must coincide."""
    import warnings
    from butd_detr_amd.bdetr import BeaUTyDETR
    from tests.golden.cases import PREFIXES, bdetr_bench_inputs
    g = load("bdetr_50k_eval.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = BeaUTyDETR(num_class=256, num_obj_class=485, input_feature_dim=3, num_queries=256,
                           num_decoder_layers=6, self_position_embedding="loc_learned",
                           contrastive_align_loss=True, butd=True, pointnet_ckpt=None, self_attend=True,
                           text_encoder_factory=text_stub.factory,
                           class_embeddings_path="/nonexistent/class_embeddings3d.npy")
    weights.fill_(model, seed=17, skip_prefixes=("text_encoder.",))
    model.cuda().eval()
    inputs = cuda(bdetr_bench_inputs())
    with torch.no_grad():
        free = model(dict(inputs))
        inputs["query_seed_inds"] = torch.from_numpy(g["query_points_sample_inds"].astype(np.int32)).cuda()
        ep = model(inputs)
    # indices: bit-exact
    np.testing.assert_array_equal(ep["sa1_inds"][:, :256].cpu().numpy(), g["sa1_inds_head"])
    np.testing.assert_array_equal(ep["sa1_inds"].long().sum(1).cpu().numpy(), g["sa1_inds_sum"])
    np.testing.assert_array_equal(ep["sa2_inds"].cpu().numpy(), g["sa2_inds"])
    assert (g["sa2_inds"] == np.arange(1024)).all()          # (the reference's own claim, backbone_module.py:131)
    np.testing.assert_array_equal(ep["seed_inds"].cpu().numpy(), g["seed_inds"])
    np.testing.assert_array_equal(ep["sa4_xyz"].cpu().numpy(), g["sa4_xyz"])
    for b in range(2):
        same = np.intersect1d(free["query_points_sample_inds"][b].cpu().numpy(), g["query_points_sample_inds"][b]).size
        assert same >= 240, same
    close(ep["seeds_obj_cls_logits"], g["seeds_obj_cls_logits"], OUT_TOL)
    close(ep["proj_tokens"], g["proj_tokens"], OUT_TOL)
    close(ep["text_memory"], g["text_memory"], OUT_TOL)
    close(ep["seed_features"][0][::4], g["seed_features_b0_rows4"], OUT_TOL)
    for pre in PREFIXES:
        close(ep[pre + "center"], g[pre + "center"], OUT_TOL)
        close(ep[pre + "pred_size"], g[pre + "pred_size"], OUT_TOL)
    close(ep["last_sem_cls_scores"][:, :, :48], g["last_sem_cls_scores_head"], OUT_TOL)
    close(ep["last_proj_queries"], g["last_proj_queries"], OUT_TOL)
    close(ep["2head_proj_queries"], g["2head_proj_queries"], OUT_TOL)


def test_bdetr_train_six_layers_golden(backend):
    """Train mode (BatchNorm batch statistics, dropout p = 0), 3 encoder + 6 decoder layers, every prefix,
    gradients from the backbone to the last decoder layer -- vs the reference (bdetr_4096_train6.npz)."""
    import warnings
    from butd_detr_amd.bdetr import BeaUTyDETR
    from tests.golden.cases import train_loss, zero_dropout
    from tests.test_golden_modules_cpu import _check_train6
    g = load("bdetr_4096_train6.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = BeaUTyDETR(num_class=256, num_obj_class=485, input_feature_dim=3, num_queries=82,
                           num_decoder_layers=6, self_position_embedding="loc_learned",
                           contrastive_align_loss=True, butd=True, pointnet_ckpt=None, self_attend=True,
                           text_encoder_factory=text_stub.factory,
                           class_embeddings_path="/nonexistent/class_embeddings3d.npy")
    weights.fill_(model, seed=15, skip_prefixes=("text_encoder.",))
    zero_dropout(model.cuda().train())
    ep = model(cuda(bdetr_inputs()))
    train_loss(ep).backward()
    # Gradients: two fp32 implementations of this train-mode model differ by more than the forward 1e-3
    # (against float64 every one of them, stock torch included, is 0.5-1.6e-2 off on the last decoder layer:
    # tests/test_gpu_gradient_truth.py, profiles/r03_gradient_error_vs_fp64.txt) --
    # the heads normalise over only 2 x 82 samples (BatchNorm1d batch statistics), whose backward subtracts
    # batch means (cancellation); stock torch on this GPU lands at 4e-3 of the scale against the CPU-run
    # reference, the fused path at 1.6e-2 on the last decoder layer in round 3 (scratch/diag_train6.py).  Round 4 (exact
    # BatchNorm sums of SA1's first layer, deterministic partial sums in the set-abstraction backward): the FUSED path is at
    # 7.8e-3 max / 1.6e-3 mean over the 14 gradient tensors, identical on every run and box seen; stock torch ops on this
    # GPU at 1.9e-2 / 1.6e-3.  Bounds: fused 1e-2 max, 2.5e-3 mean; stock ops 2.5e-2 max, 5e-3 mean.
    # Round 6 (the first layer of SA2-4 by linearity: another summation order of the same fp32 terms; per-tensor table with the
    # switch on / off and stock torch in profiles/r06_train6_gradient_errors.txt): the fused path lands at 1.36e-2 max on
    # decoder.5.ffn.3.weight and 3.4e-3 mean on decoder.5.cross_l.out_proj.bias (4.9e-3 / 1.3e-3 with the grouped-input
    # path); against a float64 run of the same modules both are inside the same flip budget
    # (tests/test_gpu_gradient_truth.py) -- which fp32 rounding the heads' 164-sample BatchNorm amplifies is not a property
    # of a kernel (stock torch: 1.9e-2 on points_obj_cls.conv2.weight).  Bounds: fused 2e-2 max / 5e-3 mean, stock ops
    # 2.5e-2 / 5e-3.
    gmax, gmean = (2e-2, 5e-3) if backend == "hip" else (2.5e-2, 5e-3)

    def grad_close(t, ref):
        a = t.detach().float().cpu().numpy()
        scale = max(float(np.abs(ref).max()), 1e-6)
        err = np.abs(a - ref) / scale
        _observe(err, gmax)
        assert err.max() <= gmax and err.mean() <= gmean, (err.max(), err.mean())

    def out_close(t, ref):
        close(t, ref, TRAIN6_OUT_TOL)

    _check_train6(ep, model, g, out_close, grad_close, backbone_tol=grad_close)


def test_bdetr_train_golden_at_the_bench_size(backend):
    """The REFERENCE model in TRAIN mode at the bench's size and batch (8 scenes x 50 000 points, 256 queries, 3 + 6 layers, dropout 0;
    golden_bdetr_bench_shape_train): forward + backward through the set-abstraction levels at their real row counts (the
    linearity paths of csrc/sa_last_bwd.hip / sa_first_linear.hip, the 10^5-row products) and every attention site at its
    real length and batch (so every kernel takes the plan it takes in the bench), against the reference's own outputs and
    gradients -- the full-size complement of the 4096-point train6 golden.  The reference's query seeds are handed back; per-query tensors are compared by seed."""
    import warnings
    from butd_detr_amd.bdetr import BeaUTyDETR
    from tests.golden.cases import PREFIXES, TRAIN_GRAD_KEYS, bdetr_bench_inputs, by_seed, train_loss, zero_dropout
    g = load("bdetr_50k_train.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = BeaUTyDETR(num_class=256, num_obj_class=485, input_feature_dim=3, num_queries=256,
                           num_decoder_layers=6, self_position_embedding="loc_learned",
                           contrastive_align_loss=True, butd=True, pointnet_ckpt=None, self_attend=True,
                           text_encoder_factory=text_stub.factory,
                           class_embeddings_path="/nonexistent/class_embeddings3d.npy")
    weights.fill_(model, seed=18, skip_prefixes=("text_encoder.",))
    zero_dropout(model.cuda().train())
    inputs = cuda(bdetr_bench_inputs(8))
    inputs["query_seed_inds"] = torch.from_numpy(g["query_seeds_sorted"].astype(np.int32)).cuda()
    ep = model(inputs)
    train_loss(ep).backward()
    np.testing.assert_array_equal(ep["seed_inds"].cpu().numpy(), g["seed_inds"])
    out_tol, gmax, gmean = BENCH_TRAIN_OUT_TOL, BENCH_TRAIN_GRAD_MAX, BENCH_TRAIN_GRAD_MEAN
    for k in ("seeds_obj_cls_logits", "proj_tokens"):
        close(ep[k], g[k], out_tol)
    close(ep["seed_features"][0][::4], g["seed_features_b0_rows4"], out_tol)
    for pre in PREFIXES:
        close(by_seed(ep, ep[pre + "center"]), g[pre + "center"], out_tol)
        close(by_seed(ep, ep[pre + "pred_size"]), g[pre + "pred_size"], out_tol)
    close(by_seed(ep, ep["last_sem_cls_scores"])[:, ::2, :32], g["last_sem_cls_scores_head"], out_tol)
    close(by_seed(ep, ep["last_proj_queries"])[:, ::2], g["last_proj_queries"], out_tol)
    close(model.backbone_net.sa2.mlp_module.layer1.bn.bn.running_mean, g["running_mean_sa2_l1"], out_tol)
    p = dict(model.named_parameters())
    worst = {}
    for k in TRAIN_GRAD_KEYS:
        gk, ref = p[k].grad, g["g_" + k]
        a = (gk[::3] if gk.dim() == 2 and gk.shape[0] >= 864 else gk).detach().float().cpu().numpy()
        err = np.abs(a - ref) / max(float(np.abs(ref).max()), 1e-6)
        _observe(err, gmax)
        worst[k] = (float(err.max()), float(err.mean()))
    bad = {k: v for k, v in worst.items() if v[0] > gmax or v[1] > gmean}
    assert not bad, bad


# ---- BASELINE configs[3]'s arithmetic ("bf16 attention / FFN") against the REFERENCE's vectors ------------------------
# The bf16 mode rounds the operands of every grouped product and of the attention core's matrix steps to bf16
# (8 bits of mantissa: 2^-9 = 2e-3 relative per operand element, fp32 accumulation).  Through a 3-layer encoder /
# a decoder layer that is a few 1e-3 of the output scale, a few 1e-2 on gradients.  Bounds (of max|ref|): the
# observed maxima on MI355X with ~2x head-room; they are what "bf16" means here, not the 1e-3 of the fp32 path.
BF16_OUT_TOL, BF16_GRAD_TOL = 2e-2, 6e-2
_BF16_OBSERVED = {}


@pytest.fixture
def bf16_mode():
    from butd_detr_amd import attention_blocks, fused_attention as fa
    prev_b = attention_blocks.get_backend()
    attention_blocks.set_backend("hip")
    prev = fa.set_compute_dtype("bf16")
    yield
    fa.set_compute_dtype(prev)
    attention_blocks.set_backend(prev_b)
    if _BF16_OBSERVED:
        root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        try:
            import json
            os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
            with open(os.path.join(root, "gpurun_out", "bf16_golden_errors.json"), "w") as f:
                json.dump(_BF16_OBSERVED, f, indent=1)
        except OSError:
            pass


def _bf16_close(case):
    def check(t, ref, tol, name):
        a = t.detach().float().cpu().numpy()
        scale = max(float(np.abs(ref).max()), 1e-6)
        err = np.abs(a - ref) / scale
        q99 = float(np.quantile(err, 0.99))
        _BF16_OBSERVED[f"{case}/{name}"] = [float(err.max()), q99, float(err.mean())]
        # 99 % of the elements within tol, the mean within tol / 4, single outliers within 4 x tol (of max|ref|)
        assert q99 <= tol and err.mean() <= tol / 4 and err.max() <= 4 * tol, \
            f"{case}/{name}: max {err.max():.3e} q99 {q99:.3e} mean {err.mean():.3e} vs {tol}"
    return check


def test_encoder_golden_bf16(bf16_mode):
    from butd_detr_amd.encoder_decoder_layers import BiEncoder, BiEncoderLayer
    g = load("encoder_small.npz")
    layer = BiEncoderLayer(288, dropout=0.1, activation="relu", n_heads=8, dim_feedforward=256,
                           self_attend_lang=True, self_attend_vis=True, use_butd_enc_attn=True)
    model = weights.fill_(BiEncoder(layer, 3), seed=11).cuda().eval()
    inp = cuda(encoder_inputs())
    for k in ("vis", "text", "pos", "boxes"):
        inp[k].requires_grad_(True)
    vis_out, text_out = model(inp["vis"], inp["pos"], inp["vis_mask"], inp["text"], inp["text_mask"],
                              {}, detected_feats=inp["boxes"], detected_mask=inp["box_mask"])
    c = _bf16_close("encoder")
    c(vis_out, g["vis_out"], BF16_OUT_TOL, "vis_out")
    c(text_out, g["text_out"], BF16_OUT_TOL, "text_out")
    ((vis_out * probe(vis_out.shape, 1).cuda()).sum() + (text_out * probe(text_out.shape, 2).cuda()).sum()).backward()
    for k, gk in (("vis", "g_vis"), ("text", "g_text"), ("pos", "g_pos"), ("boxes", "g_boxes")):
        c(inp[k].grad, g[gk], BF16_GRAD_TOL, gk)
    p = dict(model.named_parameters())
    c(p["layers.0.cross_layer.cross_lv.in_proj_weight"].grad, g["g_l0_cross_lv_in_proj_weight"], BF16_GRAD_TOL, "g_lv_in")
    c(p["layers.1.cross_layer.ffn_vl.0.weight"].grad, g["g_l1_ffn_vl_0_weight"], BF16_GRAD_TOL, "g_ffn_vl")
    c(p["layers.1.cross_layer.norm_d.weight"].grad, g["g_l1_norm_d_weight"], BF16_GRAD_TOL, "g_norm_d")


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_decoder_layer_golden_bf16(bf16_mode, mode):
    from butd_detr_amd.encoder_decoder_layers import BiDecoderLayer
    g = load(f"decoder_small_{mode}.npz")
    layer = BiDecoderLayer(288, n_heads=8, dim_feedforward=256, dropout=0.1 if mode == "eval" else 0.0,
                           activation="relu", self_position_embedding="loc_learned", butd=True)
    weights.fill_(layer, seed=12).cuda()
    layer.train(mode == "train")
    inp = cuda(decoder_inputs())
    for k in ("query", "vis", "text", "boxes"):
        inp[k].requires_grad_(True)
    out = layer(inp["query"], inp["vis"], inp["text"], inp["query_pos"], None, inp["text_mask"],
                detected_feats=inp["boxes"], detected_mask=inp["box_mask"])
    c = _bf16_close(f"decoder_{mode}")
    c(out, g["out"], BF16_OUT_TOL, "out")
    (out * probe(out.shape, 3).cuda()).sum().backward()
    for k in ("query", "vis", "text", "boxes"):
        c(inp[k].grad, g["g_" + k], BF16_GRAD_TOL, "g_" + k)
    p = dict(layer.named_parameters())
    c(p["cross_v.in_proj_weight"].grad, g["g_cross_v_in_proj_weight"], BF16_GRAD_TOL, "g_cross_v_in")
    c(p["ffn.3.weight"].grad, g["g_ffn_3_weight"], BF16_GRAD_TOL, "g_ffn_3")


def test_bdetr_train_six_layers_golden_bf16(bf16_mode):
    """configs[3]'s per-GPU arithmetic on the reference's train-mode vectors (3 + 6 layers, every prefix): indices
    bit-exact (the index ops stay fp32), outputs within the bf16 bound, gradients direction-true."""
    import warnings
    from butd_detr_amd.bdetr import BeaUTyDETR
    from tests.golden.cases import train_loss, zero_dropout
    g = load("bdetr_4096_train6.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = BeaUTyDETR(num_class=256, num_obj_class=485, input_feature_dim=3, num_queries=82,
                           num_decoder_layers=6, self_position_embedding="loc_learned",
                           contrastive_align_loss=True, butd=True, pointnet_ckpt=None, self_attend=True,
                           text_encoder_factory=text_stub.factory,
                           class_embeddings_path="/nonexistent/class_embeddings3d.npy")
    weights.fill_(model, seed=15, skip_prefixes=("text_encoder.",))
    zero_dropout(model.cuda().train())
    ep = model(cuda(bdetr_inputs()))
    train_loss(ep).backward()
    from tests.golden.cases import PREFIXES, TRAIN_GRAD_KEYS, by_seed
    np.testing.assert_array_equal(ep["seed_inds"].cpu().numpy(), g["seed_inds"])       # index ops stay fp32
    # The queries are the top-82 seeds by objectness logit: under bf16 rounding seeds near the cut may change
    # places with their neighbours, so per-query rows are compared on the seeds BOTH runs selected.
    mine = torch.sort(ep["query_points_sample_inds"].long(), dim=1)[0].cpu().numpy()
    theirs = g["query_seeds_sorted"]
    rows = []
    for b in range(mine.shape[0]):
        common = np.intersect1d(mine[b], theirs[b])
        assert len(common) >= 0.8 * mine.shape[1], (b, len(common))     # (observed: 73 and 80 of 82)
        rows.append((np.searchsorted(mine[b], common), np.searchsorted(theirs[b], common)))
    _BF16_OBSERVED["train6/common_queries"] = [float(min(len(r[0]) for r in rows)) / mine.shape[1]]

    def out_close(t, ref, name, per_query=False):
        a = t.detach().float().cpu().numpy()
        if per_query:
            a = np.concatenate([a[b][rows[b][0]] for b in range(len(rows))])
            ref = np.concatenate([ref[b][rows[b][1]] for b in range(len(rows))])
        scale = max(float(np.abs(ref).max()), 1e-6)
        err = np.abs(a - ref) / scale
        _BF16_OBSERVED[f"train6/{name}"] = [float(err.max()), float(err.mean())]
        # (through the backbone's 14 BatchNorm'd layers + 3 + 6 attention layers; the heads' BatchNorm1d over 2 x 82
        #  samples amplifies: 98 % of the elements within the bound, none beyond 5x)
        # Bounds of the bf16 mode on THIS model (train-mode BatchNorm through 14 backbone layers, 3 + 6 attention
        # layers, heads whose BatchNorm1d normalises over 2 x 82 samples; freshly initialised heads output
        # differences of large terms): feature maps and token projections -- 98 % of the elements within 6e-2 of
        # the tensor's scale; head outputs (logits, centres, sizes, query projections) -- mean error within 6e-2,
        # 99 % of the elements within 0.25, single outliers below the tensor's scale (observed: objectness logits
        # 0.15 max / 0.027 mean, proposal sizes 0.11 / 0.029, layer-4 sizes 0.56 / 0.043, centres 0.06 / 0.014).
        if per_query or name == "seeds_obj_cls_logits":
            assert err.mean() <= 6e-2 and np.quantile(err, 0.99) <= 0.25 and err.max() <= 1.0, \
                (name, err.max(), err.mean())
        else:
            assert (err > 6e-2).mean() <= 2e-2 and err.max() <= 0.3, (name, err.max(), err.mean())

    for k in ("seeds_obj_cls_logits", "proj_tokens"):
        out_close(ep[k], g[k], k)
    out_close(ep["seed_features"][0], g["seed_features_b0"], "seed_features_b0")
    out_close(by_seed(ep, ep["last_proj_queries"]), g["last_proj_queries"], "last_proj_queries", True)
    for pre in PREFIXES:
        out_close(by_seed(ep, ep[pre + "center"]), g[pre + "center"], pre + "center", True)
        out_close(by_seed(ep, ep[pre + "pred_size"]), g[pre + "pred_size"], pre + "pred_size", True)
        out_close(by_seed(ep, ep[pre + "sem_cls_scores"])[:, :, :32], g[pre + "sem_cls_scores_head"], pre + "cls", True)
    p = dict(model.named_parameters())
    cosines = {}
    for k in TRAIN_GRAD_KEYS:       # gradients: direction-true (a changed query near the cut moves them a little too)
        a, b = p[k].grad.detach().double().cpu().numpy().ravel(), g["g_" + k].astype(np.float64).ravel()
        cosines[k] = float((a * b).sum() / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-30))
        _BF16_OBSERVED[f"train6/cos:{k}"] = [cosines[k]]
    # Not only rounding: ~9 % of the 82 queries are other seeds than the reference's (see above), and the loss attaches
    # its cotangents per seed -- so a tenth of the loss terms differ.  Observed: backbone 0.79 / 0.86, decoder layers
    # 0.70 - 0.83, encoder and embeddings > 0.9.  The bound is a direction check (gross breakage gives ~0).
    bad = {k: c for k, c in cosines.items() if c <= 0.6}
    assert not bad, bad


def test_bdetr_train_six_layers_golden_bf16_same_queries(bf16_mode):
    """The parity check proper of configs[3]'s arithmetic (round 4's review: the test above is a direction check because ~9 %
    of its queries are other seeds than the reference's).  Here the one discrete choice of the forward pass is pinned:
    the model is handed the reference's 82 query seeds (``inputs["query_seed_inds"]``), so every per-query output and every
    gradient is comparable with the reference's fp32 vectors and the only difference left is bf16 rounding of the
    attention / FFN operands: head outputs within 0.4 of the tensor's scale (max) and 3.5e-2 (mean), gradient cosines >= 0.88 (observed 0.90 .. 0.998)."""
    import warnings
    from butd_detr_amd.bdetr import BeaUTyDETR
    from tests.golden.cases import PREFIXES, TRAIN_GRAD_KEYS, by_seed, train_loss, zero_dropout
    g = load("bdetr_4096_train6.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = BeaUTyDETR(num_class=256, num_obj_class=485, input_feature_dim=3, num_queries=82,
                           num_decoder_layers=6, self_position_embedding="loc_learned",
                           contrastive_align_loss=True, butd=True, pointnet_ckpt=None, self_attend=True,
                           text_encoder_factory=text_stub.factory,
                           class_embeddings_path="/nonexistent/class_embeddings3d.npy")
    weights.fill_(model, seed=15, skip_prefixes=("text_encoder.",))
    zero_dropout(model.cuda().train())
    inputs = cuda(bdetr_inputs())
    inputs["query_seed_inds"] = torch.from_numpy(g["query_seeds_sorted"].astype(np.int32)).cuda()
    ep = model(inputs)
    train_loss(ep).backward()
    np.testing.assert_array_equal(ep["seed_inds"].cpu().numpy(), g["seed_inds"])
    np.testing.assert_array_equal(torch.sort(ep["query_points_sample_inds"].long(), dim=1)[0].cpu().numpy(), g["query_seeds_sorted"])

    def close(t, ref, name):
        a = t.detach().float().cpu().numpy()
        err = np.abs(a - ref) / max(float(np.abs(ref).max()), 1e-6)
        _BF16_OBSERVED[f"train6_same/{name}"] = [float(err.max()), float(err.mean())]
        # (observed, bf16-image kernels: max 0.04 .. 0.26 -- the BatchNorm1d of a head normalises over 2 x 82 samples and
        #  amplifies single elements -- mean 0.004 .. 0.023; gpurun_out/bf16_golden_errors.json)
        assert err.max() <= 0.4 and err.mean() <= 3.5e-2, (name, float(err.max()), float(err.mean()))

    close(by_seed(ep, ep["last_proj_queries"]), g["last_proj_queries"], "last_proj_queries")
    for pre in PREFIXES:
        close(by_seed(ep, ep[pre + "center"]), g[pre + "center"], pre + "center")
        close(by_seed(ep, ep[pre + "pred_size"]), g[pre + "pred_size"], pre + "pred_size")
        close(by_seed(ep, ep[pre + "sem_cls_scores"])[:, :, :32], g[pre + "sem_cls_scores_head"], pre + "cls")
    p = dict(model.named_parameters())
    cosines = {}
    for k in TRAIN_GRAD_KEYS:
        a, b = p[k].grad.detach().double().cpu().numpy().ravel(), g["g_" + k].astype(np.float64).ravel()
        cosines[k] = float((a * b).sum() / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-30))
        _BF16_OBSERVED[f"train6_same/cos:{k}"] = [cosines[k]]
    # observed: 0.903 (SA1's first convolution, behind everything) .. 0.998; the direction check above sees 0.70 .. 0.86
    bad = {k: c for k, c in cosines.items() if c < 0.88}
    assert not bad, bad


def test_bdetr_train_golden_at_the_bench_size_bf16(bf16_mode):
    """configs[3]'s arithmetic at the bench's size and batch (8 scenes x 50 000 points, 256 queries, train mode, the reference's
    query seeds handed back) against the reference's fp32 vectors: with 8 x 256 samples per head BatchNorm -- instead of the
    4096-point golden's 2 x 82 -- the bf16 rounding of the attention / FFN operands stays a rounding error: per-query outputs
    within BF16_BENCH_OUT (max) / BF16_BENCH_OUT_MEAN (mean) of the tensor's scale, gradient cosines >= BF16_BENCH_COS."""
    import warnings
    from butd_detr_amd.bdetr import BeaUTyDETR
    from tests.golden.cases import PREFIXES, TRAIN_GRAD_KEYS, bdetr_bench_inputs, by_seed, train_loss, zero_dropout
    g = load("bdetr_50k_train.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = BeaUTyDETR(num_class=256, num_obj_class=485, input_feature_dim=3, num_queries=256,
                           num_decoder_layers=6, self_position_embedding="loc_learned",
                           contrastive_align_loss=True, butd=True, pointnet_ckpt=None, self_attend=True,
                           text_encoder_factory=text_stub.factory,
                           class_embeddings_path="/nonexistent/class_embeddings3d.npy")
    weights.fill_(model, seed=18, skip_prefixes=("text_encoder.",))
    zero_dropout(model.cuda().train())
    inputs = cuda(bdetr_bench_inputs(8))
    inputs["query_seed_inds"] = torch.from_numpy(g["query_seeds_sorted"].astype(np.int32)).cuda()
    ep = model(inputs)
    train_loss(ep).backward()
    np.testing.assert_array_equal(ep["seed_inds"].cpu().numpy(), g["seed_inds"])
    worst = [0.0, 0.0]

    def close(t, ref, name):
        a = t.detach().float().cpu().numpy()
        err = np.abs(a - ref) / max(float(np.abs(ref).max()), 1e-6)
        _BF16_OBSERVED[f"bench_size/{name}"] = [float(err.max()), float(err.mean())]
        worst[0], worst[1] = max(worst[0], float(err.max())), max(worst[1], float(err.mean()))

    for pre in PREFIXES:
        close(by_seed(ep, ep[pre + "center"]), g[pre + "center"], pre + "center")
        close(by_seed(ep, ep[pre + "pred_size"]), g[pre + "pred_size"], pre + "pred_size")
    close(by_seed(ep, ep["last_sem_cls_scores"])[:, ::2, :32], g["last_sem_cls_scores_head"], "last_cls")
    close(by_seed(ep, ep["last_proj_queries"])[:, ::2], g["last_proj_queries"], "last_proj_queries")
    p = dict(model.named_parameters())
    cosines = {}
    for k in TRAIN_GRAD_KEYS:
        gk = p[k].grad
        a = (gk[::3] if gk.dim() == 2 and gk.shape[0] >= 864 else gk).detach().double().cpu().numpy().ravel()
        b = g["g_" + k].astype(np.float64).ravel()
        cosines[k] = float((a * b).sum() / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-30))
        _BF16_OBSERVED[f"bench_size/cos:{k}"] = [cosines[k]]
    print("bf16 at the bench size: outputs max %.3e mean %.3e, gradient cosines %.4f .. %.4f" %
          (worst[0], worst[1], min(cosines.values()), max(cosines.values())))
    assert worst[0] <= BF16_BENCH_OUT and worst[1] <= BF16_BENCH_OUT_MEAN, worst
    assert min(cosines.values()) >= BF16_BENCH_COS, cosines


# observed on MI355X: outputs 7.8e-2 max / 1.1e-2 mean of the tensor's scale, gradient cosines 0.885 .. 0.999 (the 4096-point golden
# with its 2 x 82-sample BatchNorm: 0.26 / 2.3e-2, 0.90 .. 0.998); bounds ~2x / cosine 0.80
BF16_BENCH_OUT, BF16_BENCH_OUT_MEAN, BF16_BENCH_COS = 0.16, 2.5e-2, 0.80
