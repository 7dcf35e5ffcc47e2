"""Benchmark-side training step: synthetic batch, surrogate loss, optimiser, data-parallel wrap.

The reference's step is ``model(inputs) -> compute_hungarian_loss -> backward -> clip_grad_norm_(0.1)
-> AdamW.step`` (main_utils.py:415-438).  Two criteria are available:

* ``HungarianCriterion`` -- the reference's criterion (models/losses.py) as rebuilt in ``losses.py``: the
  assignment runs on the device (``butd_hungarian_match``), every term is a static-shape dense expression,
  so the whole step stays inside the hipGraph (the reference stalls 7 times per
  step on ``.cpu()`` + scipy);
* ``surrogate_loss`` -- a dense stand-in without matching that reads EVERY head output of every prefix
  (box L1, size L1, soft-token cross-entropy, query/token contrastive logits, seed objectness), kept as
  the minimal "all 21.4 M trainable parameters receive gradients" driver for kernel work.
"""
import os

import numpy as np

import torch
import torch.nn.functional as F

from . import synthetic_scenes
from .offline_text import synthetic_utterances

PREFIXES = ("proposal_", "0head_", "1head_", "2head_", "3head_", "4head_", "last_")


def synthetic_batch(batch, device, *, seed=1184, n_points=50000, tokens=80, rank=0, max_targets=16,
                    dense_cluster=False):
    """The ``inputs`` dict of train_dist_mod.py:103-110 for ``batch`` scenes (rank-dependent seeds).
    ``dense_cluster``: one piece of furniture shrunk to a quarter of its size and sampled 10x as densely
    (irregular density)."""
    base = seed + 1000 * rank
    pc = synthetic_scenes.scene_batch(batch, base, n_points, dense_cluster=dense_cluster)
    boxes, mask, cls = synthetic_scenes.detected_boxes(batch, seed=base)
    rng = np.random.default_rng(base + 99)
    targets = {
        "gt_center": torch.from_numpy(rng.uniform(-3, 3, (batch, 1, 3)).astype(np.float32)).to(device),
        "gt_size": torch.from_numpy(rng.uniform(0.3, 2, (batch, 1, 3)).astype(np.float32)).to(device),
        "gt_token": torch.from_numpy(rng.integers(1, tokens - 1, (batch,))).to(device),
        "seed_label": torch.from_numpy((rng.random((batch, 1024)) < 0.1).astype(np.float32)).to(device),
    }
    targets.update({k: torch.from_numpy(v).to(device)
                    for k, v in synthetic_ground_truth(pc, rng, tokens=tokens, max_targets=max_targets).items()})
    inputs = {
        "point_clouds": torch.from_numpy(pc).to(device),
        "text": synthetic_utterances(batch, tokens=tokens, seed=base),
        "det_boxes": torch.from_numpy(boxes).to(device),
        "det_bbox_label_mask": torch.from_numpy(mask).to(device),
        "det_class_ids": torch.from_numpy(cls).to(device),
    }
    return inputs, targets


GROUND_TRUTH_KEYS = ("center_label", "size_gts", "sem_cls_label", "positive_map", "box_label_mask",
                     "point_instance_label")


def synthetic_ground_truth(pc, rng, tokens=80, slots=132, n_class=256, max_targets=16):
    """The criterion's batch keys as joint_det_dataset.py:740-766 emits them, for synthetic scenes:
    1..max_targets (16: the grounding splits; 132: the detection split fills every slot) target boxes per scene
    centred on scene points (`center_label`, `size_gts` (B,132,3),
    `box_label_mask` (B,132)), the instance id of every point inside a target box, -1 elsewhere
    (`point_instance_label` (B,N) int64), a class id and a normalised 1-3 token span per target
    (`sem_cls_label` (B,132) int64, `positive_map` (B,132,256))."""
    B, N = pc.shape[0], pc.shape[1]
    out = {"center_label": np.zeros((B, slots, 3), np.float32), "size_gts": np.zeros((B, slots, 3), np.float32),
           "sem_cls_label": np.zeros((B, slots), np.int64), "positive_map": np.zeros((B, slots, n_class), np.float32),
           "box_label_mask": np.zeros((B, slots), np.float32),
           "point_instance_label": -np.ones((B, N), np.int64)}
    for b in range(B):
        n = int(rng.integers(1, 17)) if max_targets <= 16 else int(rng.integers(max_targets // 2, max_targets + 1))
        centres = pc[b, rng.integers(0, N, n), :3]
        sizes = rng.uniform(0.4, 1.6, (n, 3)).astype(np.float32)
        out["center_label"][b, :n], out["size_gts"][b, :n], out["box_label_mask"][b, :n] = centres, sizes, 1.0
        out["sem_cls_label"][b, :n] = rng.integers(0, 485, n)
        for t in range(n):
            start = int(rng.integers(1, max(2, tokens - 4)))
            width = int(rng.integers(1, 4))
            out["positive_map"][b, t, start:start + width] = 1.0 / width
            inside = np.all(np.abs(pc[b, :, :3] - centres[t]) <= 0.5 * sizes[t], axis=1)
            out["point_instance_label"][b, inside & (out["point_instance_label"][b] < 0)] = t
    return out


class HungarianCriterion:
    """The reference's criterion as the training loop wires it (main_utils.py:243-251, 381-387, 423-428):
    batch keys merged into ``end_points``, ``compute_hungarian_loss`` with matcher weights (1, 0, 2).
    Call ``prepare(targets)`` outside a captured region: it adds the rank-averaged box count
    (losses.py:527-534) so that no collective sits inside the graph."""

    def __init__(self, num_decoder_layers=6, use_contrastive_align=True, use_soft_token_loss=True,
                 query_points_obj_topk=4):
        from . import losses
        self._losses = losses
        names = ["boxes", "labels"] + (["contrastive_align"] if use_contrastive_align else [])
        self.set_criterion = losses.SetCriterion(
            matcher=losses.HungarianMatcher(1, 0, 2, use_soft_token_loss), losses=names, eos_coef=0.1,
            temperature=0.07)
        self.num_decoder_layers, self.topk = num_decoder_layers, query_points_obj_topk

    def prepare(self, targets):
        if "num_boxes" in targets:          # already prepared (e.g. a rank-local replay of a batch)
            return targets
        from .losses import is_dist_avail_and_initialized
        if not is_dist_avail_and_initialized():
            return targets                  # no collective to keep outside the graph: the criterion counts the boxes itself
        targets = dict(targets)
        targets["num_boxes"] = self.set_criterion.num_boxes(targets["box_label_mask"] > 0)
        return targets

    def __call__(self, end_points, targets):
        for k in GROUND_TRUTH_KEYS:
            end_points[k] = targets[k]
        if "num_boxes" in targets:
            end_points["num_boxes"] = targets["num_boxes"]
        loss, _ = self._losses.compute_hungarian_loss(end_points, self.num_decoder_layers, self.set_criterion,
                                                      self.topk)
        return loss


def surrogate_loss(end_points, targets, prefixes=None):
    """Dense differentiable loss over every head output (see module docstring)."""
    if prefixes is None:
        prefixes = [p for p in PREFIXES if f"{p}center" in end_points]
    tok = end_points["proj_tokens"]                                   # (B, L, 64)
    valid = ~end_points["text_attention_mask"]                        # (B, L) True = real token
    loss = F.binary_cross_entropy_with_logits(
        end_points["seeds_obj_cls_logits"].squeeze(1), targets["seed_label"])
    # the same four terms for every prefix: evaluated once on the stacked (P, B, Q, .) outputs
    # (sum over prefixes of per-prefix means = one sum-reduction divided by the per-prefix count)
    stack = lambda key: torch.stack([end_points[f"{p}{key}"] for p in prefixes])
    center, size = stack("center"), stack("pred_size")
    per_prefix = center[0].numel()
    loss = loss + F.smooth_l1_loss(center, targets["gt_center"].expand_as(center), reduction="sum") / per_prefix
    loss = loss + F.smooth_l1_loss(size, targets["gt_size"].expand_as(size), reduction="sum") / per_prefix
    logits = stack("sem_cls_scores")                                  # (P, B, Q, 256)
    n_p, n_b, n_q = logits.shape[:3]
    tgt = targets["gt_token"][None, :, None].expand(n_p, -1, n_q).flatten()
    loss = loss + F.cross_entropy(logits.flatten(0, 2), tgt, reduction="sum") / (n_b * n_q)
    pq = stack("proj_queries")                                        # (P, B, Q, 64)
    sim = torch.matmul(pq, tok.transpose(1, 2)) / 0.07                # (P, B, Q, L)
    sim = sim.masked_fill(~valid[None, :, None, :], -1e4)
    loss = loss + F.cross_entropy(sim.flatten(0, 2), tgt, reduction="sum") / (n_b * n_q)
    return loss


def make_optimizer(model, lr=1e-4, lr_backbone=1e-3, text_encoder_lr=1e-5, weight_decay=5e-4,
                   capturable=False):
    """AdamW with the reference's three parameter groups (main_utils.py:258-283).
    ``capturable=True`` keeps the step counters on the device so ``step()`` can sit in a hipGraph."""
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    groups = [
        {"params": [p for n, p in named if "backbone_net" not in n and "text_encoder" not in n]},
        {"params": [p for n, p in named if "backbone_net" in n], "lr": lr_backbone},
        {"params": [p for n, p in named if "text_encoder" in n], "lr": text_encoder_lr},
    ]
    groups = [g for g in groups if g["params"]]
    # capturable + fused: ONE multi-tensor kernel per parameter group; the capturable *foreach* path
    # degenerates into ~1 200 single-tensor divisions per step on this stack (rocprof, round 1)
    return torch.optim.AdamW(groups, lr=lr, weight_decay=weight_decay, capturable=capturable,
                             fused=True if capturable else None)


def train_step(model, optimizer, inputs, targets, clip_norm=0.1, criterion=None):
    """One reference-shaped iteration (main_utils.py:421-436); ``criterion(end_points, targets)`` defaults
    to the surrogate."""
    end_points = model(inputs)
    if criterion is not None and hasattr(criterion, "prepare"):
        targets = criterion.prepare(targets)
    loss = (criterion or surrogate_loss)(end_points, targets)
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


class FlatAdamW:
    """AdamW over PACKED parameters (include/butd_optim.h): the parameters of every group are moved
    into one contiguous fp32 buffer (``p.data`` become views of it, group by group, each segment padded
    to 4 floats), with matching flat gradient / moment buffers; ``step()`` is one streaming kernel per
    group.  Same update rule and groups as ``make_optimizer`` (main_utils.py:258-283); the clip
    coefficient of ``clip_grad_norm_`` is applied inside the kernel."""

    def __init__(self, model, lr=1e-4, lr_backbone=1e-3, text_encoder_lr=1e-5, weight_decay=5e-4,
                 betas=(0.9, 0.999), eps=1e-8):
        from . import _hiplib
        self._lib = _hiplib.load()
        self._check = _hiplib.check
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        # inside the first group the decoder-side parameters come first: their gradients are final when
        # backward reaches the encoder outputs, so [0, boundary_offset) of the flat buffer is the bucket an
        # overlapped exchange can send while the encoder / backbone gradients are still being computed
        pre = tuple(getattr(model.module if hasattr(model, "module") else model, "pre_boundary_prefixes", ()))
        late = lambda n: n.startswith(pre) or n.startswith(tuple("module." + q for q in pre))
        main = [(n, p) for n, p in named if "backbone_net" not in n and "text_encoder" not in n]
        first = [p for n, p in main if pre and not late(n)]
        groups = [
            (first + [p for n, p in main if not (pre and not late(n))], lr),
            ([p for n, p in named if "backbone_net" in n], lr_backbone),
            ([p for n, p in named if "text_encoder" in n], text_encoder_lr),
        ]
        groups = [(ps, l) for ps, l in groups if ps]
        self.n_first = len(first) if main else 0
        dev = named[0][1].device
        pad4 = lambda n: (n + 3) // 4 * 4
        total = sum(pad4(p.numel()) for ps, _ in groups for p in ps)
        self.flat_p = torch.zeros(total, device=dev)
        self.flat_g = torch.zeros(total, device=dev)
        self.flat_m = torch.zeros(total, device=dev)
        self.flat_v = torch.zeros(total, device=dev)
        self.param_groups, self.grad_views, self.segments = [], [], []
        off = 0
        for ps, l in groups:
            begin = off
            for p in ps:
                n = p.numel()
                self.flat_p[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat_p[off:off + n].view_as(p)
                self.grad_views.append(self.flat_g[off:off + n].view_as(p))
                off += pad4(n)
            self.segments.append((begin, off, l))
            self.param_groups.append({"params": ps, "lr": l, "weight_decay": weight_decay})
        self.params = [p for g in self.param_groups for p in g["params"]]
        # (name, offset, numel) of every parameter inside the flat buffers: a checkpoint carries it, so moments
        # saved under another ordering (e.g. another `pre_boundary_prefixes`) are re-mapped by name, never
        # silently applied to the wrong parameters
        names = {id(p): n for n, p in named}
        base = self.flat_g.data_ptr()
        self.layout = [(names[id(p)], (v.data_ptr() - base) // 4, p.numel())
                       for p, v in zip(self.params, self.grad_views)]
        # offset (in floats, a multiple of 4) where the gradients of the first `n_first` parameters end
        self.boundary_offset = ((self.grad_views[self.n_first].data_ptr() - self.flat_g.data_ptr()) // 4
                                if 0 < self.n_first < len(self.params) else 0)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.step_count = torch.zeros(1, device=dev)
        self.grad_scale = torch.ones(1, device=dev)
        self._clip_ws, self.grad_norm = None, None
        # {lr, weight_decay} per group on the device: the kernel reads them, so a captured step follows
        # whatever a scheduler writes into param_groups (main_utils.py:438 steps one every iteration)
        self.hyper = torch.zeros(len(self.param_groups), 2, device=dev)
        self._hyper_host = None
        self._hyper_sent = None
        self.sync_hyper()

    def sync_hyper(self):
        """param_groups[i]['lr' / 'weight_decay'] -> the device pair of group i (a no-op while unchanged).
        Call it before replaying a captured ``step()``; the eager ``step()`` calls it itself."""
        vals = tuple((float(g["lr"]), float(g["weight_decay"])) for g in self.param_groups)
        if vals == self._hyper_sent:
            return
        if self.hyper.is_cuda:
            if self._hyper_host is None:
                self._hyper_host = torch.empty(len(vals), 2).pin_memory()
            else:
                torch.cuda.current_stream(self.hyper.device).synchronize()   # an earlier upload may still read it
            self._hyper_host.copy_(torch.tensor(vals))
            self.hyper.copy_(self._hyper_host, non_blocking=True)
        else:
            self.hyper.copy_(torch.tensor(vals))
        self._hyper_sent = vals

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None

    def clip_(self, max_norm, grad_div=1.0):
        """clip_grad_norm_ on the flat gradient buffer: only the coefficient is computed here.  ``grad_div``:
        the buffer holds the SUM over that many ranks (the all-reduce's division is folded into the
        coefficient the kernel applies anyway: one pass over the 85.7 MB less).  On the device this is
        ``butd_clip_coefficient`` (two launches, no semaphore): ``torch.linalg.vector_norm`` of a 21 M-element
        buffer is a multi-block reduction that relies on a memset of its semaphore inside the captured graph,
        and replayed behind a still-running graph it intermittently left the norm unwritten -> coefficient 1.0
        -> an unclipped update (the round-2 "free-running" divergence)."""
        if self.flat_g.is_cuda:
            if self._clip_ws is None:
                self._clip_ws = torch.empty(int(self._lib.butd_clip_workspace_bytes()), dtype=torch.uint8,
                                            device=self.flat_g.device)
                self.grad_norm = torch.zeros(1, device=self.flat_g.device)
            err = self._lib.butd_clip_coefficient(
                self.flat_g.data_ptr(), self.flat_g.numel(), float(max_norm), float(grad_div),
                self._clip_ws.data_ptr(), self.grad_scale.data_ptr(), self.grad_norm.data_ptr(),
                torch.cuda.current_stream(self.flat_g.device).cuda_stream)
            self._check(err, "butd_clip_coefficient")
            # (the buffer is persistent: outside a capture hand back a copy, not a view later steps overwrite)
            return self.grad_norm[0] if torch.cuda.is_current_stream_capturing() else self.grad_norm[0].clone()
        norm = torch.linalg.vector_norm(self.flat_g) / grad_div
        torch.clamp(max_norm / (norm + 1e-6), max=1.0, out=self.grad_scale[0])
        if grad_div != 1.0:
            self.grad_scale.div_(grad_div)
        return norm

    def collect_grads(self):
        """Eager use (``loss.backward(); opt.step()``): the parameters' ``.grad`` tensors are not the flat
        views unless a FlatGradients packed them -- copy them in (or fail loudly when one is missing)."""
        src, dst = [], []
        for p, v in zip(self.params, self.grad_views):
            if p.grad is None:
                raise RuntimeError("FlatAdamW.step(): a trainable parameter has no gradient (the reference "
                                   "relies on every parameter receiving one, main_utils.py:310-313)")
            if p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)

    def step(self, packed=False):
        """``packed=True``: the flat gradient buffer is already filled (GraphedTrainStep); otherwise the
        parameters' ``.grad`` are gathered first."""
        if not packed:
            self.collect_grads()
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()
        self.step_count.add_(1.0)
        stream = torch.cuda.current_stream(self.flat_p.device).cuda_stream
        for i, (begin, end, lr) in enumerate(self.segments):
            err = self._lib.butd_adamw_flat(self.flat_p.data_ptr(), self.flat_g.data_ptr(),
                                            self.flat_m.data_ptr(), self.flat_v.data_ptr(), begin, end, lr,
                                            self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                            self.step_count.data_ptr(), self.grad_scale.data_ptr(),
                                            self.hyper[i].data_ptr(), stream)
            self._check(err, "butd_adamw_flat")

    # -- checkpointing (main_utils.py:144-152 saves optimizer.state_dict(), :131-141 restores it)
    def state_dict(self):
        return {"flat_m": self.flat_m.clone(), "flat_v": self.flat_v.clone(),
                "step_count": self.step_count.clone(),
                "param_groups": [{"lr": g["lr"], "weight_decay": g["weight_decay"],
                                  "numel": sum(p.numel() for p in g["params"])} for g in self.param_groups],
                "layout": [list(e) for e in self.layout],
                "betas": self.betas, "eps": self.eps}

    def load_state_dict(self, sd, allow_legacy_layout=False):
        """``allow_legacy_layout``: accept a checkpoint written before the parameter layout was recorded, ASSUMING it
        was packed in this optimizer's order (true for checkpoints of the same code version and model; a warning is
        issued because it cannot be verified)."""
        if [g["numel"] for g in sd["param_groups"]] != [sum(p.numel() for p in g["params"]) for g in self.param_groups]:
            raise ValueError("FlatAdamW.load_state_dict: parameter groups of a different model")
        theirs = [tuple(e) for e in sd.get("layout", [])]
        if not theirs:
            if not (allow_legacy_layout and sd["flat_m"].numel() == self.flat_m.numel()):
                raise ValueError("FlatAdamW.load_state_dict: the checkpoint has no parameter layout (saved before the "
                                 "layout was recorded): its moments cannot be matched to parameters safely "
                                 "(allow_legacy_layout=True assumes this optimizer's packing order)")
            import warnings
            warnings.warn("FlatAdamW.load_state_dict: layout-less checkpoint taken in the current packing order")
            theirs = list(self.layout)
        strip = lambda n: n[7:] if n.startswith("module.") else n
        if theirs == self.layout:
            self.flat_m.copy_(sd["flat_m"])
            self.flat_v.copy_(sd["flat_v"])
        else:                                            # same parameters, another ordering: re-map by name
            mine = {strip(n): (o, m) for n, o, m in self.layout}
            if sorted((strip(n), m) for n, _, m in theirs) != sorted((n, m) for n, (_, m) in mine.items()):
                raise ValueError("FlatAdamW.load_state_dict: parameter names / sizes of a different model")
            for name, off, numel in theirs:
                o, _ = mine[strip(name)]
                self.flat_m[o:o + numel].copy_(sd["flat_m"][off:off + numel])
                self.flat_v[o:o + numel].copy_(sd["flat_v"][off:off + numel])
        self.step_count.copy_(sd["step_count"])
        for g, s_g in zip(self.param_groups, sd["param_groups"]):
            g["lr"], g["weight_decay"] = s_g["lr"], s_g["weight_decay"]
        self.betas, self.eps = tuple(sd["betas"]), sd["eps"]
        self.sync_hyper()


class FlatGradients:
    """One contiguous fp32 buffer holding every trainable gradient, so the data-parallel exchange is a
    single large all-reduce (85.7 MB for the full model) instead of 601 small ones -- the message size
    RCCL's ring over the 7 xGMI links is efficient at.  ``views[i]`` aliases the
    slice of parameter i; ``gather`` fills the buffer from freshly produced ``.grad`` tensors with
    multi-tensor copies (a handful of launches), after which the parameters' ``.grad`` point at the
    views so clip/AdamW read the reduced values in place."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=torch.float32, device=ref.device)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    GATHER_CHUNK = 4096      # BUTD_GATHER_CHUNK (include/butd_optim.h)

    def gather(self, grads):
        """flat[view i] = grads[i].  On the device: ONE launch of butd_gather_segments over a pointer table
        (rebuilt only when the gradients' addresses change -- inside a captured graph they never do; the
        table's upload from pinned memory is captured with it)."""
        if not self.flat.is_cuda or any(g is None or g.dtype != torch.float32 for g in grads):
            torch._foreach_copy_(self.views, grads)
            return
        from . import _hiplib
        grads = [g if g.is_contiguous() else g.contiguous() for g in grads]
        key = tuple(g.data_ptr() for g in grads)
        n = len(grads)
        capturing = torch.cuda.is_current_stream_capturing()
        # One pointer table per CAPTURE, never rewritten: the captured upload is a node that re-reads its pinned
        # source at every replay, and several captured signatures (GraphedTrainStep's slots) replay alternately --
        # a shared table would hand slot A the gradient addresses of slot B.  Eager calls (whose gradient
        # addresses change from step to step) share one scratch table that is rewritten behind a synchronise.
        # Pinned allocations are not capturable, so every eager call leaves one spare table for the next capture.
        if getattr(self, "_gather_tables", None) is None:
            self._gather_tables, self._gather_spare, self._gather_eager = {}, None, None
            self.captured_keys = []
        make = lambda: [torch.empty(4 * n + 1, dtype=torch.int64).pin_memory(),
                        torch.empty(4 * n + 1, dtype=torch.int64, device=self.flat.device), None, 0]
        if not capturing:
            if self._gather_spare is None:
                self._gather_spare = make()
            if self._gather_eager is None:
                self._gather_eager = make()
            ent = self._gather_eager
        else:
            ent = self._gather_tables.get(key)
            if ent is None:
                if self._gather_spare is None:
                    raise RuntimeError("FlatGradients.gather: captured without an eager warm-up call before it "
                                       "(the pinned pointer table cannot be allocated during a capture)")
                ent, self._gather_spare = self._gather_spare, None
                self._gather_tables[key] = ent
                self.captured_keys.append(key)
        if ent[2] != key:
            base = self.flat.data_ptr()
            dst = [(v.data_ptr() - base) // 4 for v in self.views]
            numel = [g.numel() for g in grads]
            blk = [0]
            for m in numel:
                blk.append(blk[-1] + (m + self.GATHER_CHUNK - 1) // self.GATHER_CHUNK)
            if not capturing:
                torch.cuda.current_stream(self.flat.device).synchronize()   # an earlier upload may still read it
            ent[0].copy_(torch.tensor(list(key) + dst + numel + blk, dtype=torch.int64))
            ent[1].copy_(ent[0], non_blocking=True)
            ent[2], ent[3] = key, blk[-1]
        table, total = ent[1], ent[3]
        lib = _hiplib.load()
        with torch.cuda.device(self.flat.device):
            err = lib.butd_gather_segments(n, table.data_ptr(), self.flat.data_ptr(),
                                           torch.cuda.current_stream(self.flat.device).cuda_stream, total)
        _hiplib.check(err, "butd_gather_segments")

    def attach(self):
        for p, v in zip(self.params, self.views):
            p.grad = v

    def detach(self):
        for p in self.params:
            p.grad = None

    def all_reduce_mean(self, group=None):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(dist.get_world_size(group))

    def all_reduce_sum(self, group=None, force=False, async_op=False):
        """SUM only (the division by the world size rides on the optimizer's gradient scale); ``force`` issues
        the collective at world size 1 too (RCCL + hipGraph coexistence check on a one-GPU box)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
            return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return None


class _OptimizerGradients(FlatGradients):
    """FlatGradients over (a range of the parameters of) a FlatAdamW's own gradient buffer (no second copy)."""

    def __init__(self, opt, lo=0, hi=None):
        hi = len(opt.params) if hi is None else hi
        self.params, self.views = opt.params[lo:hi], opt.grad_views[lo:hi]
        if hi - lo == len(opt.params):
            self.flat = opt.flat_g
        else:
            begin = (self.views[0].data_ptr() - opt.flat_g.data_ptr()) // 4
            end = ((opt.grad_views[hi].data_ptr() - opt.flat_g.data_ptr()) // 4 if hi < len(opt.params)
                   else opt.flat_g.numel())
            self.flat = opt.flat_g[begin:end]


class _Slot:
    """Everything one captured signature owns: static buffers, streams, graphs."""


class GraphedTrainStep:
    """The whole iteration as hipGraph replays: tokenise on the host, copy into static buffers, replay
    ``forward_tokenized -> loss -> backward -> gather grads into the flat buffer`` (graph 1),
    all-reduce the flat buffer across ranks (outside the graph; skipped at world size 1), replay
    ``clip -> AdamW`` on the flat views (graph 2).

    ``prefetch_sampling``: the furthest-point-sampling chain of the backbone (4 serial launches, ~4.7 ms
    on 8 CUs at 8 x 50k points) depends on the coordinates only, so the graph computes it for the NEXT
    batch on a forked stream while the current batch trains on the other 248 CUs, exactly as a data
    loader prefetches: ``step(inputs_k, targets_k, next_inputs=inputs_k+1)``.  Since round 4 the branch computes the
    whole coordinate-only part of the backbone (``Pointnet2Backbone.plan``: samples, sampled centres, ball-query
    neighbour lists of the four levels, 3-NN indices + weights of the two propagation modules: ~30 launches /
    0.3 ms off the main queue); one copy node hands a batch's plan over at the start of its step.  Every replay still runs
    one sampling chain and one model pass; a call whose inputs were not announced by the previous call
    falls back to sampling them on the spot.  ``prefetch_text`` does the same with the FROZEN language
    model (its output depends on the tokens only): batch k+1's RoBERTa pass runs on a forked stream during
    batch k instead of in front of batch k+1's encoder; ignored when the text encoder is trainable.

    Data parallel (``torch.distributed`` initialised, world size N > 1; backend "nccl" is RCCL on ROCm).
    The gradient exchange is SUM all-reduces of the packed buffer; the division by N rides on the clip
    coefficient the AdamW kernel applies anyway.  With ``overlap_exchange`` (FlatAdamW + a model that exposes
    ``cut_at_encoder_output``) graph 1 is captured in two pieces: 1a = forward + loss + backward down to the
    encoder outputs, 1b = backward through the encoder and the backbone.  The decoder-side bucket (the first
    ``optimizer.boundary_offset`` floats, ~60 % of the 85.7 MB) is all-reduced asynchronously while 1b
    replays, the rest after it -- DistributedDataParallel's bucket overlap (main_utils.py:310-313) with two
    buckets and no per-parameter hooks.  Opt-in (``overlap_exchange=True`` or BUTD_OVERLAP_EXCHANGE=1): it has
    been exercised next to RCCL at world size 1 only (tests/test_gpu_free_running.py); the default is the
    single graph + one all-reduce of the whole buffer.

    Shapes are static: every (batch, points, tokens) signature is captured once and cached.  The reference
    pads the utterances to the longest of the batch (bdetr.py:160-163), and its contrastive loss takes a
    log-sum-exp over ALL token positions, padding included (losses.py:468-469) -- so the token length is part
    of the arithmetic and is kept as the reference has it by default; ``token_bucket=k`` rounds it up to a
    multiple of k (fewer captures, a marginally different contrastive term).  With N > 1 the length is the
    longest over ALL ranks (one 8-byte MAX all-reduce per step; = the reference's padding if the global
    batch sat on one GPU), so every rank captures -- and issues the warm-up collectives of -- the same
    signatures in the same order.  Warm-up steps of a capture do not train: parameters, optimizer state and
    BatchNorm buffers are restored afterwards.

    Eager PyTorch launches ~4 900 kernels per step here and is host-bound ("HIP streams and
    graphs instead of a tracing compiler"); a graph replay removes the launch overhead without
    changing a single kernel.
    """

    def __init__(self, model, optimizer, clip_norm=0.1, warmup=3, group=None, prefetch_sampling=True,
                 zero_arena=True, prefetch_text=True, criterion=None, overlap_exchange=None, token_bucket=None,
                 max_slots=None, verbose=False):
        import torch.distributed as dist
        self.model, self.optimizer, self.clip_norm, self.group = model, optimizer, clip_norm, group
        self.criterion = criterion or surrogate_loss
        self.warmup = warmup
        self.prefetch_sampling = prefetch_sampling
        self.prefetch_text = bool(prefetch_text and hasattr(self._module(), "text_encoder_is_frozen")
                                  and self._module().text_encoder_is_frozen())
        # an UtteranceCache that may serve training batches (text_stream.py): hit or miss is host control flow, so
        # the language model then runs OUTSIDE the graphs, on the text stream, under the previous step's replay
        cache = getattr(self._module(), "text_cache", None)
        self.text_outside = bool(self.prefetch_text and cache is not None and cache.cache_in_training)
        self.token_bucket = token_bucket
        self.max_slots = 8 if max_slots is None else max_slots
        self.verbose = verbose or os.environ.get("BUTD_STEP_VERBOSE", "0") == "1"
        self.step_sync = False
        # where the two prefetch branches are forked inside the captured step: "start" | "encoder" | "decoder" | "loss"
        # (measured, round 4: the language model's 2.3 ms of chip-wide stock kernels next to the launch-bound decoder,
        # 25.13 -> 24.99 ms; the sampling chain at the start)
        self.fps_fork_at, self.text_fork_at = "start", "decoder"
        self.arena = None
        from . import attention_blocks
        if zero_arena and attention_blocks.get_backend() == "hip":   # only the fused blocks draw from it
            from .fused_attention import ZeroArena
            self.arena = ZeroArena(next(model.parameters()).device)
        self.flat_opt = isinstance(optimizer, FlatAdamW)
        self.flat = (_OptimizerGradients(optimizer) if self.flat_opt else
                     FlatGradients([p for g in optimizer.param_groups for p in g["params"]]))
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        # BUTD_FORCE_COLLECTIVE=1: issue the collectives at world size 1 as well (one-GPU rehearsal of RCCL
        # next to hipGraph replays)
        self.force_collective = (os.environ.get("BUTD_FORCE_COLLECTIVE", "0") == "1"
                                 and dist.is_available() and dist.is_initialized())
        exchanging = self.world > 1 or self.force_collective
        # The two-bucket overlapped exchange has only ever run against a SELF all-reduce (one GPU, RCCL at world
        # size 1): until a run on >= 2 GPUs has compared its loss trajectory with the single-graph one it is opt-in
        # (``overlap_exchange=True`` or BUTD_OVERLAP_EXCHANGE=1); the default is one graph + one all-reduce.
        if overlap_exchange is None:
            overlap_exchange = os.environ.get("BUTD_OVERLAP_EXCHANGE", "0") == "1"
        self.split = bool(overlap_exchange and exchanging and self.flat_opt
                          and hasattr(self._module(), "cut_at_encoder_output")
                          and 0 < optimizer.n_first < len(optimizer.params))
        if self.split:
            self.flat_a = _OptimizerGradients(optimizer, 0, optimizer.n_first)
            self.flat_b = _OptimizerGradients(optimizer, optimizer.n_first, None)
        # with a process group alive its watchdog thread polls events while this thread captures: under the default
        # (global) capture mode that aborts the process (seen with RCCL at world size 1); only this thread's own
        # calls have to be capture-safe
        self._capture_mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
        self._slots, self._slot, self._sig = {}, None, None

    # -- pieces shared by the eager warm-up and the captured region
    @staticmethod
    def _plan_pieces(plan):
        """The tensors of ``Pointnet2Backbone.plan`` in a fixed order: the four sample lists first (tightly packed:
        ``inds_cur`` is their concatenation), then centres, neighbour lists, interpolation indices and weights."""
        levels, fp = plan["levels"], plan["fp"]
        return ([l[0] for l in levels] + [l[1] for l in levels] + [l[2] for l in levels]
                + [f[0] for f in fp] + [f[1] for f in fp]
                + [l[3][0] for l in levels[1:]] + [l[3][1] for l in levels[1:]])      # inverted neighbour lists, levels 2-4

    @staticmethod
    def _plan_from_pieces(pieces):
        return {"levels": [(pieces[i], pieces[4 + i], pieces[8 + i], None if i == 0 else (pieces[15 + i], pieces[18 + i]))
                           for i in range(4)],
                "fp": [(pieces[12 + i], pieces[14 + i]) for i in range(2)]}

    def _sample_into_next(self):
        """Everything of the backbone that depends on the next batch's coordinates alone (FPS chain, sampled centres,
        ball queries, 3-NN weights: Pointnet2Backbone.plan) into the `next` half of the slot's plan buffer."""
        s = self._slot
        for src, dst in zip(self._plan_pieces(self._backbone().plan(s.next_pc)), s.plan_next_views):
            dst.copy_(src)

    def _module(self):
        return self.model.module if hasattr(self.model, "module") else self.model

    def _backbone(self):
        return self._module().backbone_net

    def _encode_text_into_next(self):
        s = self._slot
        s.text_next.copy_(self._module().encode_text(s.tok_next))

    def _with_arena(self, fn, reset):
        if self.arena is None:
            return fn()
        if reset:
            self.arena.reset()                  # ONE memset for every atomics target of the step
        with self.arena:
            return fn()

    def _fwd_bwd(self):
        return self._with_arena(self._fwd_bwd_body, True)

    def _forward_loss(self):
        s = self._slot
        hooks, fired = {}, set()

        def fork(where, fn):
            """Run ``fn`` (which forks a side stream off the CURRENT point of the main stream) now, or when the forward
            pass reaches the encoder / decoder (BeaUTyDETR._stage_hook)."""
            if where not in ("start", "encoder", "decoder", "loss"):    # (an unknown stage would never fire: the branch
                raise ValueError(f"fork point {where!r}: start | encoder | decoder | loss")   # would silently not run)
            if where == "start":
                fn()
            else:
                prev = hooks.get(where)
                both = fn if prev is None else (lambda a=prev, b=fn: (a(), b()))
                hooks[where] = lambda w=where, f=both: (fired.add(w), f())

        if self.prefetch_sampling:
            s.plan_cur.copy_(s.plan_next)                    # this batch's samples / centres / neighbour lists (prefetched)

            def fork_sampling():
                s.sample_stream.wait_stream(torch.cuda.current_stream())   # fork: next batch's chain on 8 CUs
                with torch.cuda.stream(s.sample_stream):
                    self._sample_into_next()
            fork(self.fps_fork_at, fork_sampling)
        if self.prefetch_text and not self.text_outside:
            s.text_cur.copy_(s.text_next)                    # this batch's language features

            def fork_text():
                s.text_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s.text_stream):
                    self._encode_text_into_next()
            # forked where the decoder starts, not at the start of the step: the language model's 2.3 ms of chip-wide stock
            # kernels then run next to the launch-bound decoder / criterion instead of next to the HBM-bound set
            # abstraction (measured, 4 runs each of 60 steps: 25.13 -> 24.99 ms; "loss" = after the forward pass)
            fork(self.text_fork_at, fork_text)
        s.inputs["_stage_hooks"] = hooks
        try:
            end_points = self.model.forward_tokenized(s.inputs, s.tok)
        finally:
            s.inputs.pop("_stage_hooks", None)
        for where in ("encoder", "decoder", "loss"):         # ("loss", and whatever this model's forward did not call)
            if where in hooks and where not in fired:
                hooks[where]()
        return self.criterion(end_points, s.targets)

    def _join_side_streams(self):
        s = self._slot
        if self.prefetch_sampling:
            torch.cuda.current_stream().wait_stream(s.sample_stream)   # join
        if self.prefetch_text and not self.text_outside:
            torch.cuda.current_stream().wait_stream(s.text_stream)

    def _fwd_bwd_body(self):
        loss = self._forward_loss()
        self.flat.detach()                      # fresh .grad tensors: no per-parameter accumulate
        loss.backward()
        self.flat.gather([p.grad for p in self.flat.params])
        self._join_side_streams()
        return loss.detach()

    # -- the same in two pieces (overlapped exchange)
    def _stage1(self):
        def body():
            module = self._module()
            module.cut_at_encoder_output(True)
            try:
                loss = self._forward_loss()
                (outs, copies), = module._boundary
            finally:
                module.cut_at_encoder_output(False)
            self.flat.detach()
            loss.backward()
            self.flat_a.gather([p.grad for p in self.flat_a.params])
            self._slot.cut = (outs, [c.grad for c in copies])
            self._join_side_streams()           # a capture cannot end with forked work outstanding
            return loss.detach()
        return self._with_arena(body, True)

    def _stage2(self):
        def body():
            outs, grads = self._slot.cut
            self._slot.cut = None
            torch.autograd.backward(outs, grads)
            self.flat_b.gather([p.grad for p in self.flat_b.params])
        return self._with_arena(body, False)

    def _exchange_whole(self):
        if self.flat_opt:
            self.flat.all_reduce_sum(self.group, force=self.force_collective)
        else:
            self.flat.all_reduce_mean(self.group)

    def _update(self):
        if self.flat_opt:
            div = float(self.world)
            if self.clip_norm:
                self.optimizer.clip_(self.clip_norm, grad_div=div)
            else:
                self.optimizer.grad_scale.fill_(1.0 / div)
            self.optimizer.step(packed=True)
            return
        self.flat.attach()
        if self.clip_norm:
            torch.nn.utils.clip_grad_norm_(self.flat.views, self.clip_norm, foreach=True)
        self.optimizer.step()

    def _one_eager_step(self):
        if self.split:
            loss = self._stage1()
            work = self.flat_a.all_reduce_sum(self.group, force=self.force_collective, async_op=True)
            self._stage2()
            self.flat_b.all_reduce_sum(self.group, force=self.force_collective)
            if work is not None:
                work.wait()
        else:
            loss = self._fwd_bwd()
            self._exchange_whole()
        self._update()
        return loss

    def _copy_in(self, inputs, targets, tok):
        s = self._slot
        pairs = [(s.inputs[k], v) for k, v in inputs.items() if torch.is_tensor(v)]
        pairs += [(s.targets[k], v) for k, v in targets.items()]
        pairs += [(s.tok[k], tok[k]) for k in s.tok.keys()]
        self._copy_many(pairs)

    _COPY_RING = 4

    def _copy_many(self, pairs):
        """dst.copy_(src) for every pair -- the ~20 tensors a step takes over from its caller -- as ONE launch of
        butd_gather_segments over a pointer table (+ the table's upload) instead of one eager copy each: the host is
        released by the previous step's graph launch only when the GPU is almost through it, so whatever is enqueued
        eagerly between two steps is time the GPU idles (0.39 ms per step before this, profiles/r06_side_branches.txt).
        Pairs that are not 4-byte granular, not contiguous or not resident on the device take ``copy_``."""
        fast, dev = [], None
        capturing = torch.cuda.is_current_stream_capturing()
        for dst, src in pairs:
            ok = (not capturing and dst.is_cuda and src.is_cuda and src.dtype == dst.dtype and src.shape == dst.shape
                  and src.device == dst.device and src.is_contiguous() and dst.is_contiguous() and src.numel() > 0
                  and (src.numel() * src.element_size()) % 4 == 0 and src.data_ptr() % 4 == 0 and dst.data_ptr() % 4 == 0)
            if ok:
                fast.append((dst, src))
                dev = dst.device
            else:
                dst.copy_(src, non_blocking=True)
        if not fast:
            return
        import numpy as np
        from . import _hiplib
        n, chunk = len(fast), FlatGradients.GATHER_CHUNK
        ring = getattr(self, "_copy_ring", None)
        if ring is None or ring["n"] < n:
            cap = max(32, 2 * n)
            ring = self._copy_ring = {"n": cap, "i": 0, "static": {}, "slots": []}
            for _ in range(self._COPY_RING):
                host = torch.empty(4 * cap + 1, dtype=torch.int64).pin_memory()
                ring["slots"].append([host, host.numpy(), torch.empty(4 * cap + 1, dtype=torch.int64, device=dev),
                                      torch.cuda.Event()])
        # the destinations are a slot's static buffers: offsets / sizes / workgroup ranges are computed once per set
        dkey = tuple(d.data_ptr() for d, _ in fast) + tuple(d.numel() * d.element_size() for d, _ in fast)
        st = ring["static"].get(dkey)
        if st is None:
            base = min(d.data_ptr() for d, _ in fast)
            words = [d.numel() * d.element_size() // 4 for d, _ in fast]
            blk = [0]
            for w in words:
                blk.append(blk[-1] + (w + chunk - 1) // chunk)
            if len(ring["static"]) >= 64:
                ring["static"].clear()
            st = ring["static"][dkey] = (base, np.array([(d.data_ptr() - base) // 4 for d, _ in fast] + words + blk,
                                                        dtype=np.int64), blk[-1])
        base, tail, total = st
        host, host_np, table, done = ring["slots"][ring["i"] % self._COPY_RING]
        ring["i"] += 1
        done.synchronize()                                   # (the upload that last read this pinned table: long finished)
        host_np[:n] = [x.data_ptr() for _, x in fast]
        host_np[n:4 * n + 1] = tail
        stream = torch.cuda.current_stream(dev)
        table[:4 * n + 1].copy_(host[:4 * n + 1], non_blocking=True)
        done.record(stream)
        with torch.cuda.device(dev):
            err = _hiplib.load().butd_gather_segments(n, table.data_ptr(), base, stream.cuda_stream, total)
        _hiplib.check(err, "butd_gather_segments")

    # -- warm-up steps must not train
    def _snapshot(self):
        snap = {"buffers": [b.clone() for b in self.model.buffers()]}
        if self.flat_opt:
            o = self.optimizer
            snap["flat"] = [t.clone() for t in (o.flat_p, o.flat_m, o.flat_v, o.step_count)]
        else:
            import copy
            snap["params"] = [p.detach().clone() for p in self.flat.params]
            snap["opt"] = copy.deepcopy(self.optimizer.state_dict())
        return snap

    def _restore(self, snap):
        with torch.no_grad():
            for b, v in zip(self.model.buffers(), snap["buffers"]):
                b.copy_(v)
            if self.flat_opt:
                o = self.optimizer
                for t, v in zip((o.flat_p, o.flat_m, o.flat_v, o.step_count), snap["flat"]):
                    t.copy_(v)
            else:
                for p, v in zip(self.flat.params, snap["params"]):
                    p.copy_(v)
                self.optimizer.load_state_dict(snap["opt"])

    def _flats(self):
        return [f for f in (self.flat, getattr(self, "flat_a", None), getattr(self, "flat_b", None)) if f is not None]

    def _release(self, slot):
        """Drop what an evicted slot pinned outside itself: the gradient pointer tables of its captures (pinned host +
        device memory keyed by the slot's gradient addresses, FlatGradients.gather)."""
        for flat, keys in getattr(slot, "gather_keys", []):
            for key in keys:
                flat._gather_tables.pop(key, None)
        slot.gather_keys = []

    def _capture(self, inputs, targets, tok):
        from transformers import BatchEncoding
        s = self._slot = _Slot()
        seen = [(f, len(getattr(f, "captured_keys", []))) for f in self._flats()]
        s.cut = None
        s.announced = None
        s.inputs = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in inputs.items()}
        s.targets = {k: v.clone() for k, v in targets.items()}
        s.tok = BatchEncoding({k: v.clone() for k, v in tok.items()})
        if self.prefetch_sampling:
            pc = inputs["point_clouds"]
            s.next_pc = pc[..., :3].clone()
            # one flat buffer per half (`next`: written by the prefetch branch, `cur`: read by this step's backbone) for
            # everything Pointnet2Backbone.plan produces; ONE copy node hands a batch's plan over at the start of its step
            pieces = self._plan_pieces(self._backbone().plan(s.next_pc))    # (THIS batch: shapes + first contents)
            offs, o = [], 0
            for i, t in enumerate(pieces):
                if i >= 4:
                    o = (o + 255) // 256 * 256                # (the four sample lists stay contiguous: inds_cur below)
                offs.append(o)
                o += t.numel() * t.element_size()
            s.plan_next = torch.empty(o, dtype=torch.uint8, device=pc.device)
            s.plan_cur = torch.empty_like(s.plan_next)
            view = lambda buf: [buf[off:off + t.numel() * t.element_size()].view(t.dtype).view(t.shape)
                                for off, t in zip(offs, pieces)]
            s.plan_next_views, cur_views = view(s.plan_next), view(s.plan_cur)
            s.inputs["backbone_plan"] = self._plan_from_pieces(cur_views)
            n_inds = sum(t.numel() for t in pieces[:4])
            s.inds_cur = s.plan_cur[:4 * n_inds].view(torch.int32)    # the concatenated sample lists of the current batch
            s.sample_stream = self._own_stream("sample")
            for src, dst in zip(pieces, s.plan_next_views):   # prime with THIS batch
                dst.copy_(src)
            torch.cuda.synchronize()
        if self.prefetch_text:
            s.tok_next = BatchEncoding({k: v.clone() for k, v in tok.items()})
            s.text_next = self._module().encode_text(s.tok_next, inputs.get("text")).clone()   # THIS batch
            s.text_cur = s.text_next.clone()
            s.inputs["text_encoder_output"] = s.text_cur
            s.text_stream = self._own_stream("text")
            torch.cuda.synchronize()
        snap = self._snapshot()
        side = self._own_stream("warmup")
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self._one_eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._restore(snap)
        if self.arena is not None and not self.arena.captured:
            self.arena.reset()           # (eager: folds the chunks the warm-up steps discovered into one before the
        #                                   first capture bakes their addresses in -- one zero fill per step, not one per chunk)
        torch.cuda.synchronize()
        # graphs keep their hipGraph_t until the first replay: memset nodes (hipMemsetAsync of a stock torch op, e.g.
        # the semaphore reset of a multi-block reduction) are rewritten into kernel nodes first -- replayed memset
        # nodes are unreliable on ROCm 7.2 (graph_audit.py)
        from . import graph_audit
        s.graphs = {}
        cap = self._own_stream("capture")     # (torch's default capture stream is a pool stream too)
        if self.split:
            s.g_stage1 = s.graphs["stage1"] = graph_audit.new_graph()
            with torch.cuda.graph(s.g_stage1, capture_error_mode=self._capture_mode, stream=cap):
                s.loss = self._stage1()
            s.g_stage2 = s.graphs["stage2"] = graph_audit.new_graph()
            with torch.cuda.graph(s.g_stage2, pool=s.g_stage1.pool(), capture_error_mode=self._capture_mode, stream=cap):
                self._stage2()
            pool = s.g_stage1.pool()
        else:
            s.g_fwd_bwd = s.graphs["fwd_bwd"] = graph_audit.new_graph()
            with torch.cuda.graph(s.g_fwd_bwd, capture_error_mode=self._capture_mode, stream=cap):
                s.loss = self._fwd_bwd()
            pool = s.g_fwd_bwd.pool()
        s.g_update = s.graphs["update"] = graph_audit.new_graph()
        with torch.cuda.graph(s.g_update, pool=pool, capture_error_mode=self._capture_mode, stream=cap):
            self._update()
        s.memset_nodes_rewritten = {k: graph_audit.make_safe(g) for k, g in s.graphs.items()}
        s.node_inventory = {k: dict(graph_audit.inventory(g)) for k, g in s.graphs.items()}
        for g in s.graphs.values():
            g.instantiate()
        s.gather_keys = [(f, list(getattr(f, "captured_keys", [])[n0:])) for f, n0 in seen]
        # the captures above ran the optimizer once more under capture semantics only (nothing executed)

    def _pad_tokens(self, tok):
        """Pad (input_ids with the pad id 1, attention_mask with 0) to the bucketed / rank-agreed length."""
        length = int(tok["input_ids"].shape[1])
        want = length
        if self.token_bucket:
            want = (length + self.token_bucket - 1) // self.token_bucket * self.token_bucket
        if self.world > 1 or self.force_collective:   # (forced at world size 1: the one-GPU rehearsal of this path)
            import torch.distributed as dist
            dev = tok["input_ids"].device
            if dist.get_backend(self.group) == "gloo" or dev.type != "cuda":
                t = torch.tensor([want], dtype=torch.int64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            else:
                # on the upload stream: the collective (and the .item() behind it) must not queue behind the
                # previous step's graphs on the main stream -- the host would stall until that step has finished
                # and the GPU would idle while the next graphs are being enqueued
                with torch.cuda.stream(self._upload_stream(dev)):
                    t = torch.tensor([want], dtype=torch.int64, device=dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                    t = t.cpu()                      # (the read-back is stream-ordered too: inside the context)
            want = int(t.item())
        if want == length:
            return tok
        from transformers import BatchEncoding
        out = {}
        for k, v in tok.items():
            fill = 1 if k == "input_ids" else 0
            out[k] = torch.nn.functional.pad(v, (0, want - length), value=fill)
        return BatchEncoding(out)

    def _own_stream(self, name, dev=None):
        """The step's streams are its own HIP streams, one per role and shared by all captured signatures -- never members
        of torch's pool of 32, which RCCL's stream is drawn from as well (graph_audit.own_stream)."""
        # (LOW-priority streams for the prefetch branches were measured in round 6: 26.1 instead of 20.5 ms per step -- a graph
        # with nodes of another priority leaves the runtime's fast path, as round 4 saw for a high-priority launch stream;
        # profiles/r06_side_branches.txt)
        from . import graph_audit
        return graph_audit.own_stream(dev, role="step." + name, owner=self)

    def _upload_stream(self, dev):
        if getattr(self, "_up", None) is None:
            self._up = self._own_stream("upload", dev)
            self._up_done = torch.cuda.Event()
            self._up_pinned = {}
        return self._up

    def _tokenize(self, inputs):
        """Host tokenisation (bdetr.py:164-166) + upload.  ``BatchEncoding.to(device)`` from pageable memory is a
        synchronous copy on the current stream: the host waited there until the previous step's graphs had finished
        and only then enqueued the next ones (0.8 ms of idle GPU per step in the trace).  The ids go through pinned
        staging buffers on an upload stream instead; the main stream waits for the upload's event."""
        module = self._module()
        dev = inputs["point_clouds"].device
        if dev.type != "cuda" or not hasattr(module, "tokenizer"):
            return self._pad_tokens(self.model.tokenize(inputs))
        from transformers import BatchEncoding
        # (the tokeniser is a pure function of the utterances: a batch seen before -- every epoch revisits the same
        # utterances -- costs a dictionary look-up instead of 0.2 ms of host time in front of the step's graph launch)
        memo = self.__dict__.setdefault("_tok_memo", {})
        key = (dev.index,) + tuple(inputs["text"])
        out = memo.get(key)
        if out is None:
            host = module.tokenizer.batch_encode_plus(inputs["text"], padding="longest", return_tensors="pt")
            up = self._upload_stream(dev)
            self._up_done.synchronize()                  # the previous upload has left its staging buffers
            out = {}
            with torch.cuda.stream(up):
                for k, v in host.items():
                    pk = (k, tuple(v.shape), v.dtype)
                    pin = self._up_pinned.get(pk)
                    if pin is None:
                        pin = self._up_pinned[pk] = torch.empty_like(v).pin_memory()
                    pin.copy_(v)
                    out[k] = pin.to(dev, non_blocking=True)
                self._up_done.record(up)
            main = torch.cuda.current_stream(dev)
            main.wait_event(self._up_done)
            for v in out.values():
                v.record_stream(main)
            if len(memo) >= 4096:
                memo.pop(next(iter(memo)))
            memo[key] = out          # (the ids stay resident: 8 x 80 int64 x 2 per batch of utterances)
        return self._pad_tokens(BatchEncoding(out))

    def __call__(self, inputs, targets, next_inputs=None):
        # step_sync (debug hook, tests/test_gpu_free_running.py): wait for the previous step before enqueueing this one.
        # Round 2 needed it for the two-piece step; the cause was a replayed MEMSET node (torch's vector_norm semaphore)
        # -- gone since the clip coefficient is butd_clip_coefficient and captured graphs are scrubbed of memset nodes.
        if self.step_sync:
            torch.cuda.current_stream().synchronize()
        # host work stays in the step; a batch announced by the previous call was tokenised then
        cache = getattr(self, "_tok_cache", None)
        tok = cache[1] if (cache is not None and cache[0] is inputs) else self._tokenize(inputs)
        self._tok_cache = None
        if hasattr(self.criterion, "prepare"):                 # collectives of the criterion: outside the graph
            targets = self.criterion.prepare(targets)
        sig = (tuple(inputs["point_clouds"].shape), tuple(tok["input_ids"].shape))
        if sig != self._sig:
            if sig in self._slots:
                self._slot = self._slots.pop(sig)              # (re-inserted below: most recently used last)
                self._slot.announced = None
            else:
                # a slot owns graphs, a private memory pool and static buffers (several GB at 8 x 50 000 points):
                # least recently used signatures are dropped first so a dataset with many token lengths cannot
                # exhaust HBM (max_slots; `token_bucket` keeps the number of signatures small in the first place)
                while len(self._slots) >= max(1, self.max_slots):
                    old_sig = next(iter(self._slots))
                    torch.cuda.synchronize()                   # its last replay may still run
                    self._release(self._slots.pop(old_sig))
                self._slot, self._sig = None, None             # (a failed capture must not leave a stale signature)
                self._capture(inputs, targets, tok)
                self._slot.announced = inputs
                if self.verbose:
                    print(f"[GraphedTrainStep] captured signature {sig}: {len(self._slots) + 1} slot(s), "
                          f"{torch.cuda.memory_reserved() / 2**30:.1f} GiB reserved; nodes {self._slot.node_inventory}; "
                          f"memset nodes rewritten {self._slot.memset_nodes_rewritten}", flush=True)
            self._slots[sig] = self._slot
            self._sig = sig
        s = self._slot
        self._copy_in(inputs, targets, tok)
        announced = s.announced is inputs
        nxt = next_inputs if next_inputs is not None else inputs
        if self.prefetch_sampling:
            if not announced:                                  # not prefetched: sample it now
                s.next_pc.copy_(inputs["point_clouds"][..., :3])
                self._sample_into_next()
            s.next_pc.copy_(nxt["point_clouds"][..., :3], non_blocking=True)
        text_ok = True
        if self.prefetch_text and self.text_outside:
            main, module = torch.cuda.current_stream(), self._module()
            if announced:
                main.wait_stream(s.text_stream)                # encoded (or gathered) under the previous replay
            else:
                s.text_next.copy_(module.encode_text(tok, inputs.get("text")))
            s.text_cur.copy_(s.text_next)
            tok_next = tok if nxt is inputs else self._tokenize(nxt)
            text_ok = tuple(tok_next["input_ids"].shape) == tuple(s.tok_next["input_ids"].shape)
            if text_ok:
                s.text_stream.wait_stream(main)
                with torch.cuda.stream(s.text_stream):         # cached utterances: one row gather, no RoBERTa
                    s.text_next.copy_(module.encode_text(tok_next, nxt.get("text")))
            if nxt is not inputs:
                self._tok_cache = (nxt, tok_next)
        elif self.prefetch_text:
            if not announced:                                  # not prefetched: encode it now
                for k in s.tok_next.keys():
                    s.tok_next[k].copy_(tok[k])
                self._encode_text_into_next()
            tok_next = tok if nxt is inputs else self._tokenize(nxt)
            text_ok = tuple(tok_next["input_ids"].shape) == tuple(s.tok_next["input_ids"].shape)
            if text_ok:
                for k in s.tok_next.keys():
                    s.tok_next[k].copy_(tok_next[k], non_blocking=True)
            if nxt is not inputs:
                self._tok_cache = (nxt, tok_next)
        # a next batch of another token length runs in another slot: treat it as unannounced there
        s.announced = nxt if (next_inputs is not None and text_ok) else None
        if self.flat_opt:
            self.optimizer.sync_hyper()                        # a scheduler may have moved the learning rates
        if self.split:
            s.g_stage1.replay()
            work = self.flat_a.all_reduce_sum(self.group, force=self.force_collective, async_op=True)
            s.g_stage2.replay()                                # encoder + backbone backward under the exchange
            self.flat_b.all_reduce_sum(self.group, force=self.force_collective)
            if work is not None:
                work.wait()
        else:
            s.g_fwd_bwd.replay()
            self._exchange_whole()
        s.g_update.replay()
        return s.loss
