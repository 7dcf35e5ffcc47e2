"""Generates tests/golden/*.npz by IMPORTING THE REFERENCE (/root/reference) in this container.

    python tests/golden/make_golden.py

Runs only where /root/reference exists (never on the GPU box); the committed .npz files are the
fixtures, this script is their provenance.  What runs here is the reference's own Python:
models/encoder_decoder_layers.py, models/backbone_module.py + pointnet2/*.py, models/bdetr.py, on CPU
fp32 (torch 2.10), with four shims that change no arithmetic on the path being pinned:

  * ``pointnet2._ext``      -> oracle/ext_adapter.py (the CPU oracle; the reference has no CPU ops)
  * RoBERTa + tokenizer     -> tests/golden/text_stub.py (parameter-free closed form; no weights offline)
  * ``models`` package init -> bypassed (it imports wandb/plyfile/ipdb-dependent eval code)
  * weights                 -> tests/golden/weights.py fill_ (name-keyed, reproducible on both sides)

Inputs are regenerated from seeds by the tests (see ``inputs_*`` below), so the .npz files hold
expected outputs (+ the few inputs that are cheaper to store than to describe).
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

from golden import text_stub, weights  # noqa: E402
from golden.cases import (PREFIXES, TRAIN_GRAD_KEYS, backbone_inputs, bdetr_bench_inputs, bdetr_inputs, by_seed,  # noqa: E402
                          decoder_bench_inputs, decoder_inputs, encoder_inputs, probe, train_loss, zero_dropout)


def load_reference():
    """Import the reference modules with the shims described in the module docstring."""
    from oracle import ext_adapter as oracle_ext
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "pointnet2"))
    sys.modules["pointnet2._ext"] = oracle_ext
    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(REF, "models")]
    sys.modules["models"] = pkg
    enc = importlib.import_module("models.encoder_decoder_layers")
    bb = importlib.import_module("models.backbone_module")
    bdetr = importlib.import_module("models.bdetr")
    return enc, bb, bdetr


def npz(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB")


def golden_encoder(enc):
    torch.manual_seed(0)
    layer = enc.BiEncoderLayer(288, dropout=0.1, activation="relu", n_heads=8, dim_feedforward=256,
                               self_attend_lang=True, self_attend_vis=True, use_butd_enc_attn=True)
    model = weights.fill_(enc.BiEncoder(layer, 3), seed=11).eval()
    inp = encoder_inputs()
    for k in ("vis", "text", "pos", "boxes"):
        inp[k].requires_grad_(True)
    vis_out, text_out = model(inp["vis"], inp["pos"], inp["vis_mask"], inp["text"], inp["text_mask"],
                              {}, detected_feats=inp["boxes"], detected_mask=inp["box_mask"])
    loss = (vis_out * probe(vis_out.shape, 1)).sum() + (text_out * probe(text_out.shape, 2)).sum()
    loss.backward()
    p = dict(model.named_parameters())
    npz("encoder_small.npz", vis_out=vis_out, text_out=text_out,
        g_vis=inp["vis"].grad, g_text=inp["text"].grad, g_pos=inp["pos"].grad,
        g_boxes=inp["boxes"].grad,
        g_l0_cross_lv_in_proj_weight=p["layers.0.cross_layer.cross_lv.in_proj_weight"].grad,
        g_l2_self_attention_visual_out_proj_weight=p["layers.2.self_attention_visual.self_attn.out_proj.weight"].grad,
        g_l1_ffn_vl_0_weight=p["layers.1.cross_layer.ffn_vl.0.weight"].grad,
        g_l1_norm_d_weight=p["layers.1.cross_layer.norm_d.weight"].grad)


def golden_decoder(enc):
    for mode in ("eval", "train"):
        # train mode: dropout=0 so the only train/eval difference is BatchNorm1d batch statistics
        layer = enc.BiDecoderLayer(288, n_heads=8, dim_feedforward=256,
                                   dropout=0.1 if mode == "eval" else 0.0, activation="relu",
                                   self_position_embedding="loc_learned", butd=True)
        weights.fill_(layer, seed=12)
        layer.train(mode == "train")
        inp = decoder_inputs()
        for k in ("query", "vis", "text", "boxes"):
            inp[k].requires_grad_(True)
        out = layer(inp["query"], inp["vis"], inp["text"], inp["query_pos"], None, inp["text_mask"],
                    detected_feats=inp["boxes"], detected_mask=inp["box_mask"])
        (out * probe(out.shape, 3)).sum().backward()
        p = dict(layer.named_parameters())
        npz(f"decoder_small_{mode}.npz", out=out, g_query=inp["query"].grad, g_vis=inp["vis"].grad,
            g_text=inp["text"].grad, g_boxes=inp["boxes"].grad,
            g_cross_v_in_proj_weight=p["cross_v.in_proj_weight"].grad,
            g_self_posembed_0_weight=p["self_posembed.position_embedding_head.0.weight"].grad,
            g_ffn_3_weight=p["ffn.3.weight"].grad)


def golden_decoder_bench_shape(enc):
    """One decoder layer at the bench's shapes (cases.decoder_bench_inputs), train mode, dropout 0: the sites
    256 x 256, 256 x 80, 256 x 132, 256 x 1024 that the one-pass attention backward serves in the step."""
    layer = enc.BiDecoderLayer(288, n_heads=8, dim_feedforward=256, dropout=0.0, activation="relu",
                               self_position_embedding="loc_learned", butd=True)
    weights.fill_(layer, seed=16)
    layer.train()
    inp = decoder_bench_inputs()
    for k in ("query", "vis", "text", "boxes"):
        inp[k].requires_grad_(True)
    out = layer(inp["query"], inp["vis"], inp["text"], inp["query_pos"], None, inp["text_mask"],
                detected_feats=inp["boxes"], detected_mask=inp["box_mask"])
    (out * probe(out.shape, 5)).sum().backward()
    p = dict(layer.named_parameters())
    # (the fixture stays small: every 2nd query row, every 8th seed row + column sums, every 3rd row of the packed
    # q | k | v weight gradients)
    npz("decoder_256x1024_train.npz", out_rows2=out[:, ::2], g_query_rows2=inp["query"].grad[:, ::2],
        g_vis_rows8=inp["vis"].grad[:, ::8], g_vis_colsum=inp["vis"].grad.double().sum(1),
        g_text=inp["text"].grad, g_boxes_rows2=inp["boxes"].grad[:, ::2],
        g_cross_v_in_proj_weight_rows3=p["cross_v.in_proj_weight"].grad[::3],
        g_self_attn_in_proj_weight_rows3=p["self_attn.in_proj_weight"].grad[::3],
        g_cross_d_out_proj_weight=p["cross_d.out_proj.weight"].grad,
        g_self_posembed_0_weight=p["self_posembed.position_embedding_head.0.weight"].grad,
        g_ffn_3_weight=p["ffn.3.weight"].grad)


def golden_backbone(bb):
    for mode in ("eval", "train"):
        net = weights.fill_(bb.Pointnet2Backbone(input_feature_dim=3, width=1), seed=13)
        net.train(mode == "train")
        pc = backbone_inputs()
        ep = net(pc, end_points={})
        loss = (ep["fp2_features"] * probe(ep["fp2_features"].shape, 4)).sum()
        loss.backward()
        p = dict(net.named_parameters())
        npz(f"backbone_4096_{mode}.npz",
            sa1_inds=ep["sa1_inds"], sa2_inds=ep["sa2_inds"], fp2_inds=ep["fp2_inds"],
            sa1_xyz=ep["sa1_xyz"], sa4_xyz=ep["sa4_xyz"],
            sa1_features_head=ep["sa1_features"][:, :, :64], sa2_features_head=ep["sa2_features"][:, :, :64],
            sa4_features=ep["sa4_features"], fp2_features_b0=ep["fp2_features"][0],
            fp2_features_sum=ep["fp2_features"].double().sum(), fp2_features_abs_sum=ep["fp2_features"].double().abs().sum(),
            g_sa1_layer0_conv=p["sa1.mlp_module.layer0.conv.weight"].grad,
            g_sa3_layer2_conv=p["sa3.mlp_module.layer2.conv.weight"].grad,
            g_sa2_layer1_bn_weight=p["sa2.mlp_module.layer1.bn.bn.weight"].grad,
            g_fp1_layer0_conv=p["fp1.mlp.layer0.conv.weight"].grad,
            g_fp2_layer1_conv=p["fp2.mlp.layer1.conv.weight"].grad)


def golden_bdetr(bdetr):
    tok, txt = text_stub.factory()
    bdetr.RobertaTokenizerFast = types.SimpleNamespace(from_pretrained=lambda *_a, **_k: tok)
    bdetr.RobertaModel = types.SimpleNamespace(from_pretrained=lambda *_a, **_k: txt)
    cwd = os.getcwd()
    os.chdir(REF)  # bdetr.py:88-91 loads data/class_embeddings3d.npy relative to cwd
    try:
        model = bdetr.BeaUTyDETR(num_class=256, num_obj_class=485, input_feature_dim=3,
                                 num_queries=32, num_decoder_layers=2,
                                 self_position_embedding="loc_learned", contrastive_align_loss=True,
                                 butd=True, pointnet_ckpt=None, self_attend=True)
    finally:
        os.chdir(cwd)
    keys = sorted(k for k in model.state_dict() if not k.startswith("text_encoder."))
    weights.fill_(model, seed=14, skip_prefixes=("text_encoder.",))
    model.eval()
    inputs = bdetr_inputs()
    with torch.no_grad():
        ep = model(inputs)
    out = {k: ep[k] for k in (
        "seed_inds", "seeds_obj_cls_logits", "query_points_sample_inds", "text_feats", "text_memory",
        "proj_tokens", "proposal_center", "proposal_pred_size", "proposal_proj_queries",
        "0head_center", "last_center", "last_pred_size", "last_sem_cls_scores", "last_proj_queries",
        "text_attention_mask")}
    out["seed_features_b0"] = ep["seed_features"][0]
    out["end_points_keys"] = np.array(sorted(k for k in ep.keys()))
    out["state_dict_keys"] = np.array(keys)
    out["state_dict_shapes"] = np.array([str(tuple(model.state_dict()[k].shape)) for k in keys])
    npz("bdetr_4096_eval.npz", **out)


def golden_bdetr_bench_shape(bdetr):
    """The reference model at the bench's size -- 2 scenes x 50 000 points, 256 queries, 3 encoder + 6 decoder layers --
    in eval mode: the sampled indices of levels 1 and 2, the centres of level 4, the query seeds (the tests hand them
    back as ``inputs["query_seed_inds"]``: two objectness logits within rounding distance may otherwise swap a seed in or
    out of the top 256), and every prefix's outputs.  Index ops: the CPU oracle (ext_adapter), as in every golden."""
    tok, txt = text_stub.factory()
    bdetr.RobertaTokenizerFast = types.SimpleNamespace(from_pretrained=lambda *_a, **_k: tok)
    bdetr.RobertaModel = types.SimpleNamespace(from_pretrained=lambda *_a, **_k: txt)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        model = bdetr.BeaUTyDETR(num_class=256, num_obj_class=485, input_feature_dim=3,
                                 num_queries=256, num_decoder_layers=6,
                                 self_position_embedding="loc_learned", contrastive_align_loss=True,
                                 butd=True, pointnet_ckpt=None, self_attend=True)
    finally:
        os.chdir(cwd)
    weights.fill_(model, seed=17, skip_prefixes=("text_encoder.",))
    model.eval()
    with torch.no_grad():
        ep = model(bdetr_bench_inputs())
    out = {"sa1_inds_head": ep["sa1_inds"][:, :256], "sa1_inds_sum": ep["sa1_inds"].long().sum(1),
           "sa2_inds": ep["sa2_inds"], "sa4_xyz": ep["sa4_xyz"], "seed_inds": ep["seed_inds"],
           "query_points_sample_inds": ep["query_points_sample_inds"],
           "seeds_obj_cls_logits": ep["seeds_obj_cls_logits"], "proj_tokens": ep["proj_tokens"],
           "text_memory": ep["text_memory"], "seed_features_b0_rows4": ep["seed_features"][0][::4]}
    for pre in PREFIXES:
        out[pre + "center"] = ep[pre + "center"]
        out[pre + "pred_size"] = ep[pre + "pred_size"]
    out["last_sem_cls_scores_head"] = ep["last_sem_cls_scores"][:, :, :48]
    out["last_proj_queries"] = ep["last_proj_queries"]
    out["2head_proj_queries"] = ep["2head_proj_queries"]
    npz("bdetr_50k_eval.npz", **out)


def golden_bdetr_train(bdetr):
    """BeaUTyDETR in TRAIN mode (BatchNorm batch statistics, dropout p = 0), all 6 decoder layers,
    forward + gradients -- the config-2 architecture on a 4096-point cloud."""
    tok, txt = text_stub.factory()
    bdetr.RobertaTokenizerFast = types.SimpleNamespace(from_pretrained=lambda *_a, **_k: tok)
    bdetr.RobertaModel = types.SimpleNamespace(from_pretrained=lambda *_a, **_k: txt)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        model = bdetr.BeaUTyDETR(num_class=256, num_obj_class=485, input_feature_dim=3,
                                 num_queries=82, num_decoder_layers=6,
                                 self_position_embedding="loc_learned", contrastive_align_loss=True,
                                 butd=True, pointnet_ckpt=None, self_attend=True)
    finally:
        os.chdir(cwd)
    weights.fill_(model, seed=15, skip_prefixes=("text_encoder.",))
    model.train()
    zero_dropout(model)
    ep = model(bdetr_inputs())
    train_loss(ep).backward()
    out = {"seed_inds": ep["seed_inds"],
           "query_seeds_sorted": torch.sort(ep["query_points_sample_inds"].long(), dim=1)[0],
           "seeds_obj_cls_logits": ep["seeds_obj_cls_logits"], "proj_tokens": ep["proj_tokens"],
           "seed_features_b0": ep["seed_features"][0]}
    for pre in PREFIXES:
        out[pre + "center"] = by_seed(ep, ep[pre + "center"])
        out[pre + "pred_size"] = by_seed(ep, ep[pre + "pred_size"])
        out[pre + "sem_cls_scores_head"] = by_seed(ep, ep[pre + "sem_cls_scores"])[:, :, :32]
    out["last_proj_queries"] = by_seed(ep, ep["last_proj_queries"])
    p = dict(model.named_parameters())
    missing = [n for n, q in p.items() if q.requires_grad and q.grad is None]
    assert not missing, missing            # DDP without find_unused_parameters relies on this
    for k in TRAIN_GRAD_KEYS:
        out["g_" + k] = p[k].grad
    out["running_mean_sa2_l1"] = model.backbone_net.sa2.mlp_module.layer1.bn.bn.running_mean
    npz("bdetr_4096_train6.npz", **out)


def golden_bdetr_bench_shape_train(bdetr):
    """The reference model at the bench's size AND batch in TRAIN mode (BatchNorm batch statistics, dropout p = 0): 8 scenes x
    50 000 points, 256 queries, 3 + 6 layers, forward + backward of tests.golden.cases.train_loss -- the set-abstraction levels at
    their real row counts (2 x 2048 x 64 ... groups), every attention site at its real length.  Per-query tensors by seed;
    large tensors stored as every 4th / 3rd row."""
    tok, txt = text_stub.factory()
    bdetr.RobertaTokenizerFast = types.SimpleNamespace(from_pretrained=lambda *_a, **_k: tok)
    bdetr.RobertaModel = types.SimpleNamespace(from_pretrained=lambda *_a, **_k: txt)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        model = bdetr.BeaUTyDETR(num_class=256, num_obj_class=485, input_feature_dim=3,
                                 num_queries=256, num_decoder_layers=6,
                                 self_position_embedding="loc_learned", contrastive_align_loss=True,
                                 butd=True, pointnet_ckpt=None, self_attend=True)
    finally:
        os.chdir(cwd)
    weights.fill_(model, seed=18, skip_prefixes=("text_encoder.",))
    model.train()
    zero_dropout(model)
    ep = model(bdetr_bench_inputs(8))
    train_loss(ep).backward()
    out = {"seed_inds": ep["seed_inds"].to(torch.int32),
           "query_seeds_sorted": torch.sort(ep["query_points_sample_inds"].long(), dim=1)[0].to(torch.int32),
           "seeds_obj_cls_logits": ep["seeds_obj_cls_logits"], "proj_tokens": ep["proj_tokens"],
           "seed_features_b0_rows4": ep["seed_features"][0][::4]}
    for pre in PREFIXES:
        out[pre + "center"] = by_seed(ep, ep[pre + "center"])
        out[pre + "pred_size"] = by_seed(ep, ep[pre + "pred_size"])
    out["last_sem_cls_scores_head"] = by_seed(ep, ep["last_sem_cls_scores"])[:, ::2, :32]
    out["last_proj_queries"] = by_seed(ep, ep["last_proj_queries"])[:, ::2]
    p = dict(model.named_parameters())
    assert not [n for n, q in p.items() if q.requires_grad and q.grad is None]
    for k in TRAIN_GRAD_KEYS:
        gk = p[k].grad
        out["g_" + k] = gk[::3] if gk.dim() == 2 and gk.shape[0] >= 864 else gk
    out["running_mean_sa2_l1"] = model.backbone_net.sa2.mlp_module.layer1.bn.bn.running_mean
    npz("bdetr_50k_train.npz", **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    enc_m, bb_m, bdetr_m = load_reference()
    if sys.argv[1:] == ["decoder_bench_shape"]:       # (one case alone: the others are unchanged)
        golden_decoder_bench_shape(enc_m)
        sys.exit(0)
    if sys.argv[1:] == ["bdetr_bench_shape"]:
        golden_bdetr_bench_shape(bdetr_m)
        sys.exit(0)
    if sys.argv[1:] == ["bdetr_bench_shape_train"]:
        golden_bdetr_bench_shape_train(bdetr_m)
        sys.exit(0)
    golden_encoder(enc_m)
    golden_decoder(enc_m)
    golden_decoder_bench_shape(enc_m)
    golden_backbone(bb_m)
    golden_bdetr(bdetr_m)
    golden_bdetr_bench_shape(bdetr_m)
    golden_bdetr_train(bdetr_m)
    golden_bdetr_bench_shape_train(bdetr_m)
