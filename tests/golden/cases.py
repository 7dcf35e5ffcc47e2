"""Seeded inputs of the golden cases -- imported by make_golden.py (reference side) and by the tests
(build side), so both feed identical tensors."""
import numpy as np
import torch

D_MODEL = 288


def _t(rng, *shape, scale=1.0):
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))


def probe(shape, seed):
    """Fixed random cotangent: loss = sum(out * probe) makes backward deterministic and dense."""
    rng = np.random.default_rng(1000 + seed)
    return torch.from_numpy(rng.standard_normal(tuple(shape)).astype(np.float32))


def encoder_inputs():
    rng = np.random.default_rng(21)
    b, v, l, d = 2, 64, 12, 10
    text_mask = torch.zeros(b, l, dtype=torch.bool)
    text_mask[1, 8:] = True
    box_mask = torch.zeros(b, d, dtype=torch.bool)
    box_mask[0, 6:] = True
    box_mask[1, 3:] = True
    return dict(vis=_t(rng, b, v, D_MODEL), pos=_t(rng, b, v, D_MODEL, scale=0.5),
                text=_t(rng, b, l, D_MODEL), boxes=_t(rng, b, d, D_MODEL),
                vis_mask=torch.zeros(b, v, dtype=torch.bool), text_mask=text_mask, box_mask=box_mask)


def decoder_inputs():
    rng = np.random.default_rng(22)
    b, q, v, l, d = 2, 16, 64, 12, 10
    text_mask = torch.zeros(b, l, dtype=torch.bool)
    text_mask[0, 9:] = True
    box_mask = torch.zeros(b, d, dtype=torch.bool)
    box_mask[1, 4:] = True
    return dict(query=_t(rng, b, q, D_MODEL), vis=_t(rng, b, v, D_MODEL), text=_t(rng, b, l, D_MODEL),
                boxes=_t(rng, b, d, D_MODEL), query_pos=_t(rng, b, q, 6),
                text_mask=text_mask, box_mask=box_mask)


def decoder_bench_inputs():
    """A decoder layer at the bench's per-layer shapes (256 queries, 1024 seeds, 80 tokens, 132 boxes), 6 scenes: the
    batch from which the attention backward's one-pass kernel takes its 256-key plan at the 1024-seed site, as at
    the bench's 8 (csrc/attention_ops.hip longk_plan: (Lk / 256) * B * heads >= 192)."""
    rng = np.random.default_rng(23)
    b, q, v, l, d = 6, 256, 1024, 80, 132
    text_mask = torch.zeros(b, l, dtype=torch.bool)
    box_mask = torch.zeros(b, d, dtype=torch.bool)
    for i in range(b):
        text_mask[i, 12 + 9 * i:] = True            # 12 .. 57 tokens
        box_mask[i, 20 + 17 * i:] = True            # 20 .. 105 boxes
    return dict(query=_t(rng, b, q, D_MODEL), vis=_t(rng, b, v, D_MODEL), text=_t(rng, b, l, D_MODEL),
                boxes=_t(rng, b, d, D_MODEL), query_pos=_t(rng, b, q, 6),
                text_mask=text_mask, box_mask=box_mask)


def backbone_inputs():
    from butd_detr_amd.synthetic_scenes import uniform_cloud
    return torch.from_numpy(uniform_cloud(seed=5, n_points=4096, batch=2))


def bdetr_inputs():
    from butd_detr_amd.synthetic_scenes import detected_boxes, uniform_cloud
    pc = torch.from_numpy(uniform_cloud(seed=6, n_points=4096, batch=2))
    boxes, mask, cls = detected_boxes(2, seed=6, min_valid=5, max_valid=20)
    return {"point_clouds": pc,
            "text": ["find the chair that is next to the brown wooden table",
                     "the lamp on the desk"],
            "det_boxes": torch.from_numpy(boxes), "det_bbox_label_mask": torch.from_numpy(mask),
            "det_class_ids": torch.from_numpy(cls)}


_BENCH_TEXTS = ["the office chair between the desk and the window that faces the door",
                "find the trash can under the table next to the whiteboard",
                "the brown wooden cabinet left of the sink",
                "a monitor on the second desk from the entrance",
                "the couch facing the television",
                "it is the lamp standing in the corner behind the armchair",
                "the smaller of the two bookshelves near the bed",
                "select the door that is closest to the refrigerator"]


def bdetr_bench_inputs(batch=2):
    """``batch`` scenes at the bench's size: 50 000 ScanNet-shaped points each, 132 box slots, utterances of the text stub
    (2: the eval golden; 8: the train golden -- the bench's own batch, so every kernel takes the plan it takes there)."""
    from butd_detr_amd.synthetic_scenes import detected_boxes, scene_batch
    pc = torch.from_numpy(np.ascontiguousarray(scene_batch(batch, 4242, 50000)))
    boxes, mask, cls = detected_boxes(batch, seed=7, min_valid=20, max_valid=90)
    return {"point_clouds": pc, "text": list(_BENCH_TEXTS[:batch]),
            "det_boxes": torch.from_numpy(boxes), "det_bbox_label_mask": torch.from_numpy(mask),
            "det_class_ids": torch.from_numpy(cls)}


# ---- train-mode BeaUTyDETR golden (bdetr_4096_train6.npz): shared by make_golden.py and the tests ----
TRAIN_GRAD_KEYS = (
    "backbone_net.sa1.mlp_module.layer0.conv.weight", "backbone_net.fp2.mlp.layer1.conv.weight",
    "cross_encoder.layers.0.self_attention_visual.self_attn.in_proj_weight",
    "cross_encoder.layers.2.cross_layer.cross_d.out_proj.weight",
    "cross_encoder.layers.1.cross_layer.ffn_lv.0.weight",
    "decoder.0.self_attn.in_proj_weight", "decoder.3.cross_v.in_proj_weight", "decoder.5.ffn.3.weight",
    "decoder.5.cross_l.out_proj.bias", "prediction_heads.4.center_residual_head.net.0.weight",
    "proposal_head.size_pred_head.net.8.weight", "contrastive_align_projection_image.0.weight",
    "text_projector.0.weight", "points_obj_cls.conv2.weight")
PREFIXES = ("proposal_", "0head_", "1head_", "2head_", "3head_", "4head_", "last_")


def zero_dropout(model):
    """Train mode without randomness: every Dropout p = 0 (the attention modules keep their own p)."""
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(getattr(m, "dropout", None), float):
            m.dropout = 0.0
    return model


def by_seed(ep, t):
    """Rows of a per-query tensor (B, Q, C) re-ordered by the query's seed index: the model orders its queries
    by objectness (top-k), and two logits within rounding distance of each other may swap places between two
    fp32 implementations; the decoder is permutation-equivariant over queries, so everything compared or
    differentiated in this case is expressed per SEED, not per position."""
    order = torch.argsort(ep["query_points_sample_inds"].long(), dim=1)
    return torch.gather(t, 1, order[..., None].expand(-1, -1, t.shape[-1]))


def train_loss(ep):
    """Dense scalar touching every head of every prefix: fixed random cotangents attached to the seed a
    query sits on (see by_seed), so the loss and its gradients do not depend on the order of the queries."""
    dev = ep["seeds_obj_cls_logits"].device
    inds = ep["query_points_sample_inds"].long()
    loss = (ep["seeds_obj_cls_logits"] * probe(ep["seeds_obj_cls_logits"].shape, 40).to(dev)).sum()
    loss = loss + (ep["proj_tokens"] * probe(ep["proj_tokens"].shape, 41).to(dev)).sum()
    n_seed = ep["seeds_obj_cls_logits"].shape[-1]
    for i, pre in enumerate(PREFIXES):
        for j, k in enumerate(("center", "pred_size", "sem_cls_scores", "proj_queries")):
            t = ep[pre + k]
            per_seed = probe((t.shape[0], n_seed, t.shape[-1]), 50 + 4 * i + j).to(dev)
            cot = torch.gather(per_seed, 1, inds[..., None].expand(-1, -1, t.shape[-1]))
            loss = loss + (t * cot).sum()
    return loss
