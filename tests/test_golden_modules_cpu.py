"""CPU (-m "not gpu"): the build's host-side modules vs golden vectors captured from the REFERENCE
(tests/golden/make_golden.py).  The pointnet2 operator backend is monkeypatched to the oracle here --
in tests only -- because the product operators have no CPU path.  Tolerance: north_star's 1e-3 fp32
on outputs (these CPU-vs-CPU runs land around 1e-5), bit-exact on indices."""
import os

import numpy as np
import pytest
import torch

from tests import oracle_ext
from tests.golden import text_stub, weights
from tests.golden.cases import (backbone_inputs, bdetr_inputs, decoder_inputs, encoder_inputs,
                                probe)

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = dict(rtol=1e-3, atol=1e-3)


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def close(t, ref, **kw):
    np.testing.assert_allclose(t.detach().cpu().numpy(), ref, **(kw or TOL))


def grad_close(t, ref):
    scale = max(float(np.abs(ref).max()), 1e-6)
    np.testing.assert_allclose(t.detach().cpu().numpy() / scale, ref / scale, rtol=0, atol=2e-3)


@pytest.fixture()
def cpu_ops(monkeypatch):
    from butd_detr_amd import pointnet2_utils
    monkeypatch.setattr(pointnet2_utils, "_ext", oracle_ext)


def test_encoder_matches_reference_golden():
    from butd_detr_amd.encoder_decoder_layers import BiEncoder, BiEncoderLayer
    g = load("encoder_small.npz")
    layer = BiEncoderLayer(288, dropout=0.1, activation="relu", n_heads=8, dim_feedforward=256,
                           self_attend_lang=True, self_attend_vis=True, use_butd_enc_attn=True)
    model = weights.fill_(BiEncoder(layer, 3), seed=11).eval()
    inp = encoder_inputs()
    for k in ("vis", "text", "pos", "boxes"):
        inp[k].requires_grad_(True)
    vis_out, text_out = model(inp["vis"], inp["pos"], inp["vis_mask"], inp["text"], inp["text_mask"],
                              {}, detected_feats=inp["boxes"], detected_mask=inp["box_mask"])
    close(vis_out, g["vis_out"])
    close(text_out, g["text_out"])
    ((vis_out * probe(vis_out.shape, 1)).sum() + (text_out * probe(text_out.shape, 2)).sum()).backward()
    grad_close(inp["vis"].grad, g["g_vis"])
    grad_close(inp["text"].grad, g["g_text"])
    grad_close(inp["pos"].grad, g["g_pos"])
    grad_close(inp["boxes"].grad, g["g_boxes"])
    p = dict(model.named_parameters())
    grad_close(p["layers.0.cross_layer.cross_lv.in_proj_weight"].grad, g["g_l0_cross_lv_in_proj_weight"])
    grad_close(p["layers.2.self_attention_visual.self_attn.out_proj.weight"].grad,
               g["g_l2_self_attention_visual_out_proj_weight"])
    grad_close(p["layers.1.cross_layer.ffn_vl.0.weight"].grad, g["g_l1_ffn_vl_0_weight"])
    grad_close(p["layers.1.cross_layer.norm_d.weight"].grad, g["g_l1_norm_d_weight"])


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_decoder_layer_matches_reference_golden(mode):
    from butd_detr_amd.encoder_decoder_layers import BiDecoderLayer
    g = load(f"decoder_small_{mode}.npz")
    layer = BiDecoderLayer(288, n_heads=8, dim_feedforward=256,
                           dropout=0.1 if mode == "eval" else 0.0, activation="relu",
                           self_position_embedding="loc_learned", butd=True)
    weights.fill_(layer, seed=12)
    layer.train(mode == "train")
    inp = decoder_inputs()
    for k in ("query", "vis", "text", "boxes"):
        inp[k].requires_grad_(True)
    out = layer(inp["query"], inp["vis"], inp["text"], inp["query_pos"], None, inp["text_mask"],
                detected_feats=inp["boxes"], detected_mask=inp["box_mask"])
    close(out, g["out"])
    (out * probe(out.shape, 3)).sum().backward()
    grad_close(inp["query"].grad, g["g_query"])
    grad_close(inp["vis"].grad, g["g_vis"])
    grad_close(inp["text"].grad, g["g_text"])
    grad_close(inp["boxes"].grad, g["g_boxes"])
    p = dict(layer.named_parameters())
    grad_close(p["cross_v.in_proj_weight"].grad, g["g_cross_v_in_proj_weight"])
    grad_close(p["self_posembed.position_embedding_head.0.weight"].grad, g["g_self_posembed_0_weight"])
    grad_close(p["ffn.3.weight"].grad, g["g_ffn_3_weight"])


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_backbone_matches_reference_golden(cpu_ops, mode):
    from butd_detr_amd.backbone_module import Pointnet2Backbone
    g = load(f"backbone_4096_{mode}.npz")
    net = weights.fill_(Pointnet2Backbone(input_feature_dim=3, width=1), seed=13)
    net.train(mode == "train")
    ep = net(backbone_inputs(), end_points={})
    for k in ("sa1_inds", "sa2_inds", "fp2_inds"):
        np.testing.assert_array_equal(ep[k].numpy(), g[k])
    close(ep["sa1_xyz"], g["sa1_xyz"], rtol=0, atol=0)
    close(ep["sa4_xyz"], g["sa4_xyz"], rtol=0, atol=0)
    close(ep["sa1_features"][:, :, :64], g["sa1_features_head"])
    close(ep["sa2_features"][:, :, :64], g["sa2_features_head"])
    close(ep["sa4_features"], g["sa4_features"])
    close(ep["fp2_features"][0], g["fp2_features_b0"])
    assert abs(float(ep["fp2_features"].double().sum()) - float(g["fp2_features_sum"])) < 1e-3 * float(g["fp2_features_abs_sum"])
    (ep["fp2_features"] * probe(ep["fp2_features"].shape, 4)).sum().backward()
    p = dict(net.named_parameters())
    grad_close(p["sa1.mlp_module.layer0.conv.weight"].grad, g["g_sa1_layer0_conv"])
    grad_close(p["sa3.mlp_module.layer2.conv.weight"].grad, g["g_sa3_layer2_conv"])
    grad_close(p["sa2.mlp_module.layer1.bn.bn.weight"].grad, g["g_sa2_layer1_bn_weight"])
    grad_close(p["fp1.mlp.layer0.conv.weight"].grad, g["g_fp1_layer0_conv"])
    grad_close(p["fp2.mlp.layer1.conv.weight"].grad, g["g_fp2_layer1_conv"])


def test_bdetr_state_dict_keys_and_forward_match_reference_golden(cpu_ops):
    from butd_detr_amd.bdetr import BeaUTyDETR
    g = load("bdetr_4096_eval.npz")
    with pytest.warns(UserWarning, match="class_embeddings3d"):
        model = BeaUTyDETR(num_class=256, num_obj_class=485, input_feature_dim=3, num_queries=32,
                           num_decoder_layers=2, self_position_embedding="loc_learned",
                           contrastive_align_loss=True, butd=True, pointnet_ckpt=None,
                           self_attend=True, text_encoder_factory=text_stub.factory,
                           class_embeddings_path="/nonexistent/class_embeddings3d.npy")
    sd = model.state_dict()
    keys = sorted(k for k in sd if not k.startswith("text_encoder."))
    assert keys == list(g["state_dict_keys"]), "state_dict key set differs from the reference"
    assert [str(tuple(sd[k].shape)) for k in keys] == list(g["state_dict_shapes"])
    weights.fill_(model, seed=14, skip_prefixes=("text_encoder.",))
    model.eval()
    with torch.no_grad():
        ep = model(bdetr_inputs())
    assert sorted(ep.keys()) == list(g["end_points_keys"])
    for k in ("seed_inds", "query_points_sample_inds", "text_attention_mask"):
        np.testing.assert_array_equal(ep[k].numpy(), g[k])
    for k in ("seeds_obj_cls_logits", "text_feats", "text_memory", "proj_tokens", "proposal_center",
              "proposal_pred_size", "proposal_proj_queries", "0head_center", "last_center",
              "last_pred_size", "last_sem_cls_scores", "last_proj_queries"):
        close(ep[k], g[k])
    close(ep["seed_features"][0], g["seed_features_b0"])
    # every trainable parameter gets a gradient (DDP without find_unused_parameters, main_utils.py:310-313)
    model.train()
    ep = model(bdetr_inputs())
    loss = sum(ep[k].float().pow(2).mean() for k in ep
               if torch.is_tensor(ep[k]) and ep[k].is_floating_point() and ep[k].requires_grad)
    loss.backward()
    missing = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
    assert missing == []
    # optimizer grouping of main_utils.py:258-280 relies on these substrings
    names = [n for n, _ in model.named_parameters()]
    assert any("backbone_net" in n for n in names)


def _check_train6(ep, model, g, out_close, g_close, backbone_tol=None):
    """Outputs of all 7 prefixes + gradients vs bdetr_4096_train6.npz (reference, train mode, dropout 0)."""
    from tests.golden.cases import PREFIXES, TRAIN_GRAD_KEYS, by_seed
    np.testing.assert_array_equal(ep["seed_inds"].cpu().numpy(), g["seed_inds"])
    # the same SET of seeds becomes queries (their order may differ by swaps of near-equal logits: by_seed)
    np.testing.assert_array_equal(torch.sort(ep["query_points_sample_inds"].long(), dim=1)[0].cpu().numpy(),
                                  g["query_seeds_sorted"])
    for k in ("seeds_obj_cls_logits", "proj_tokens"):
        out_close(ep[k], g[k])
    out_close(by_seed(ep, ep["last_proj_queries"]), g["last_proj_queries"])
    out_close(ep["seed_features"][0], g["seed_features_b0"])
    for pre in PREFIXES:
        out_close(by_seed(ep, ep[pre + "center"]), g[pre + "center"])
        out_close(by_seed(ep, ep[pre + "pred_size"]), g[pre + "pred_size"])
        out_close(by_seed(ep, ep[pre + "sem_cls_scores"])[:, :, :32], g[pre + "sem_cls_scores_head"])
    p = dict(model.named_parameters())
    for k in TRAIN_GRAD_KEYS:
        if backbone_tol is not None and k.startswith("backbone_net."):
            backbone_tol(p[k].grad, g["g_" + k])
        else:
            g_close(p[k].grad, g["g_" + k])
    out_close(model.backbone_net.sa2.mlp_module.layer1.bn.bn.running_mean, g["running_mean_sa2_l1"])


def test_bdetr_train_mode_six_decoder_layers_matches_reference_golden(cpu_ops):
    """The config-2 architecture (3 encoder + 6 decoder layers) in TRAIN mode: BatchNorm batch statistics,
    every prefix's heads, gradients from the backbone to the last decoder layer."""
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
    zero_dropout(model.train())
    ep = model(bdetr_inputs())
    train_loss(ep).backward()
    _check_train6(ep, model, g, close, grad_close)


def test_bdetr_at_the_bench_size_matches_reference_golden(cpu_ops):
    """The host-side mirror at the bench's size (2 scenes x 50 000 points, 256 queries, 3 + 6 layers, eval mode) on the
    CPU with the oracle's index ops, against the reference's vectors (bdetr_50k_eval.npz): what the GPU test of the same
    name checks for the HIP path, here for the module code alone."""
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
    model.eval()
    with torch.no_grad():
        ep = model(bdetr_bench_inputs())
    for k in ("sa2_inds", "seed_inds", "query_points_sample_inds"):
        np.testing.assert_array_equal(ep[k].cpu().numpy(), g[k])
    np.testing.assert_array_equal(ep["sa1_inds"][:, :256].cpu().numpy(), g["sa1_inds_head"])
    for pre in PREFIXES:
        close(ep[pre + "center"], g[pre + "center"])
        close(ep[pre + "pred_size"], g[pre + "pred_size"])
    close(ep["last_proj_queries"], g["last_proj_queries"])
    close(ep["seeds_obj_cls_logits"], g["seeds_obj_cls_logits"])
