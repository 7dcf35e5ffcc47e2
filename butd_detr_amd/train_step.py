"""Benchmark-side training step: synthetic batch, surrogate loss, optimiser, data-parallel wrap.

The reference's step is ``model(inputs) -> compute_hungarian_loss -> backward -> clip_grad_norm_(0.1)
-> AdamW.step`` (main_utils.py:415-438).  The Hungarian criterion (models/losses.py, scipy on the
host, 7 D2H syncs per step) is outside the hot path this repo builds, so
the timed step uses a dense on-device surrogate that reads EVERY head output of every prefix --
box L1, size L1, soft-token cross-entropy, query/token contrastive logits, seed objectness -- so all
21.4 M trainable parameters receive gradients exactly as in the reference step (DDP without
``find_unused_parameters``), followed by the same clip + AdamW update.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import synthetic_scenes
from .offline_text import synthetic_utterances

PREFIXES = ("proposal_", "0head_", "1head_", "2head_", "3head_", "4head_", "last_")


def synthetic_batch(batch, device, *, seed=1184, n_points=50000, tokens=80, rank=0):
    """The ``inputs`` dict of train_dist_mod.py:103-110 for ``batch`` scenes (rank-dependent seeds)."""
    base = seed + 1000 * rank
    pc = synthetic_scenes.scene_batch(batch, base, n_points)
    boxes, mask, cls = synthetic_scenes.detected_boxes(batch, seed=base)
    rng = np.random.default_rng(base + 99)
    targets = {
        "gt_center": torch.from_numpy(rng.uniform(-3, 3, (batch, 1, 3)).astype(np.float32)).to(device),
        "gt_size": torch.from_numpy(rng.uniform(0.3, 2, (batch, 1, 3)).astype(np.float32)).to(device),
        "gt_token": torch.from_numpy(rng.integers(1, tokens - 1, (batch,))).to(device),
        "seed_label": torch.from_numpy((rng.random((batch, 1024)) < 0.1).astype(np.float32)).to(device),
    }
    inputs = {
        "point_clouds": torch.from_numpy(pc).to(device),
        "text": synthetic_utterances(batch, tokens=tokens, seed=base),
        "det_boxes": torch.from_numpy(boxes).to(device),
        "det_bbox_label_mask": torch.from_numpy(mask).to(device),
        "det_class_ids": torch.from_numpy(cls).to(device),
    }
    return inputs, targets


def surrogate_loss(end_points, targets, prefixes=None):
    """Dense differentiable loss over every head output (see module docstring)."""
    if prefixes is None:
        prefixes = [p for p in PREFIXES if f"{p}center" in end_points]
    tok = end_points["proj_tokens"]                                   # (B, L, 64)
    valid = ~end_points["text_attention_mask"]                        # (B, L) True = real token
    loss = F.binary_cross_entropy_with_logits(
        end_points["seeds_obj_cls_logits"].squeeze(1), targets["seed_label"])
    for p in prefixes:
        center, size = end_points[f"{p}center"], end_points[f"{p}pred_size"]
        loss = loss + F.smooth_l1_loss(center, targets["gt_center"].expand_as(center))
        loss = loss + F.smooth_l1_loss(size, targets["gt_size"].expand_as(size))
        logits = end_points[f"{p}sem_cls_scores"]                      # (B, Q, 256)
        tgt = targets["gt_token"][:, None].expand(-1, logits.shape[1])
        loss = loss + F.cross_entropy(logits.flatten(0, 1), tgt.flatten())
        pq = end_points[f"{p}proj_queries"]                            # (B, Q, 64)
        sim = torch.matmul(pq, tok.transpose(1, 2)) / 0.07             # (B, Q, L)
        sim = sim.masked_fill(~valid[:, None, :], -1e4)
        loss = loss + F.cross_entropy(sim.flatten(0, 1), tgt.flatten())
    return loss


def make_optimizer(model, lr=1e-4, lr_backbone=1e-3, text_encoder_lr=1e-5, weight_decay=5e-4):
    """AdamW with the reference's three parameter groups (main_utils.py:258-283)."""
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    groups = [
        {"params": [p for n, p in named if "backbone_net" not in n and "text_encoder" not in n]},
        {"params": [p for n, p in named if "backbone_net" in n], "lr": lr_backbone},
        {"params": [p for n, p in named if "text_encoder" in n], "lr": text_encoder_lr},
    ]
    groups = [g for g in groups if g["params"]]
    return torch.optim.AdamW(groups, lr=lr, weight_decay=weight_decay)


def train_step(model, optimizer, inputs, targets, clip_norm=0.1):
    """One reference-shaped iteration (main_utils.py:421-436) with the surrogate criterion."""
    end_points = model(inputs)
    loss = surrogate_loss(end_points, targets)
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    if clip_norm:
        params = [p for g in optimizer.param_groups for p in g["params"]]
        torch.nn.utils.clip_grad_norm_(params, clip_norm)
    optimizer.step()
    return loss


def wrap_data_parallel(model, device):
    """DistributedDataParallel exactly as main_utils.py:310-313 (``broadcast_buffers=False``, no
    SyncBN, no find_unused_parameters).  The gradient all-reduce (85.7 MB fp32) is the path's only
    collective; backend 'nccl' on ROCm is RCCL over xGMI."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model
    if device.type == "cuda":
        return DistributedDataParallel(model, device_ids=[device.index], broadcast_buffers=False,
                                       gradient_as_bucket_view=True)
    return DistributedDataParallel(model, broadcast_buffers=False)
