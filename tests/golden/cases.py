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
