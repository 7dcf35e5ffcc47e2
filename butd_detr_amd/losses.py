"""The reference's training criterion (models/losses.py) without its host round trips.

Reference flow (losses.py:546-617, 516-543, 257-331): for each of the 7 prefixes (proposal, 5 intermediate
heads, last) build the (B*Q, sum n_b) cost matrix, move it to the host (`.cpu()`: a device sync), run
scipy's linear_sum_assignment once per scene, move the index lists back, gather/scatter with them
(data-dependent shapes), and `.item()` the all-reduced box count: 7 stalls per step that no hipGraph can
hold.  This module computes the same numbers with STATIC shapes and no synchronisation:

* targets stay in their padded (B, G) slots with `box_label_mask` as a validity mask (ascending slot order
  = the reference's compacted order);
* the cost tensors of all prefixes are built at once in the (P, B, G, Q) orientation (the transpose scipy
  solves internally) and every (prefix, scene) problem is solved on the device by one wavefront
  (`butd_hungarian_match`, include/butd_lsap.h) -> `match (P, B, G)`: query of every valid slot, -1 else;
* the four loss terms are masked dense expressions of `match` evaluated once on the stacked (P, B, Q, .)
  head outputs; the box count stays a device scalar (all-reduced as a tensor under torch.distributed).

Public names mirror the reference so `train_dist_mod.py`'s calls read the same: `HungarianMatcher`
(:226), `SetCriterion` (:334), `compute_hungarian_loss` (:546), `compute_points_obj_cls_loss_hard_topk`
(:161), `generalized_box_iou3d` (:70), `box_cxcyczwhd_to_xyzxyz` (:27), `SigmoidFocalClassificationLoss`
(:94).  The list-of-dict `targets` API of the reference is kept on `HungarianMatcher.forward` /
`SetCriterion.forward` (it has to build Python lists, so it synchronises); the dense entry points are what
the step uses.  The assignment has no CPU fallback: without the HIP library the matcher raises.
"""
import torch
import torch.distributed as dist
import torch.nn as nn

from . import _hiplib

def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


# ------------------------------------------------------------------------------------------ boxes
def box_cxcyczwhd_to_xyzxyz(x):
    """(..., 6) centre + size -> (..., 6) min / max corners, sizes clamped at 1e-6 (losses.py:27-37)."""
    c, s = x[..., :3], torch.clamp(x[..., 3:], min=1e-6)
    return torch.cat([c - 0.5 * s, c + 0.5 * s], dim=-1)


def _volume(b):
    return (b[..., 3] - b[..., 0]) * (b[..., 4] - b[..., 1]) * (b[..., 5] - b[..., 2])


def _giou_broadcast(a, b):
    """Generalised IoU of corner boxes with broadcasting leading dims (losses.py:40-91 arithmetic)."""
    lo = torch.maximum(a[..., :3], b[..., :3])
    hi = torch.minimum(a[..., 3:], b[..., 3:])
    edge = torch.clamp(hi - lo, min=0)
    inter = edge[..., 0] * edge[..., 1] * edge[..., 2]
    union = _volume(a) + _volume(b) - inter
    iou = inter / union
    hull = torch.clamp(torch.maximum(a[..., 3:], b[..., 3:]) - torch.minimum(a[..., :3], b[..., :3]), min=0)
    vol = hull[..., 0] * hull[..., 1] * hull[..., 2]
    return iou - (vol - union) / vol


def _volume_par(box):
    """(N, 6) xyzxyz boxes -> (N,) volumes (models/losses.py:40-45)."""
    return (box[:, 3] - box[:, 0]) * (box[:, 4] - box[:, 1]) * (box[:, 5] - box[:, 2])


def _intersect_par(box_a, box_b):
    """Pairwise intersection volumes (N, M) (models/losses.py:48-59)."""
    lo = torch.max(box_a[:, None, :3], box_b[None, :, :3])
    hi = torch.min(box_a[:, None, 3:], box_b[None, :, 3:])
    return (hi - lo).clamp(min=0).prod(-1)


def _iou3d_par(box_a, box_b):
    """Pairwise 3-D IoU and union, as src/grounding_evaluator.py:11 imports it (models/losses.py:62-67)."""
    inter = _intersect_par(box_a, box_b)
    union = _volume_par(box_a)[:, None] + _volume_par(box_b)[None, :] - inter
    return inter / union, union


def generalized_box_iou3d(boxes1, boxes2):
    """(N, 6), (M, 6) corner boxes -> (N, M) pairwise GIoU (losses.py:70-91)."""
    return _giou_broadcast(boxes1[:, None, :], boxes2[None, :, :])


# ------------------------------------------------------------------------------------------ focal
class SigmoidFocalClassificationLoss(nn.Module):
    """losses.py:94-158 (Group-Free's sigmoid focal loss)."""

    def __init__(self, gamma=2.0, alpha=0.25):
        super().__init__()
        self.alpha, self.gamma = alpha, gamma

    @staticmethod
    def sigmoid_cross_entropy_with_logits(input, target):
        return torch.clamp(input, min=0) - input * target + torch.log1p(torch.exp(-torch.abs(input)))

    def forward(self, input, target, weights):
        p = torch.sigmoid(input)
        alpha_weight = target * self.alpha + (1 - target) * (1 - self.alpha)
        pt = target * (1.0 - p) + (1.0 - target) * p
        loss = alpha_weight * torch.pow(pt, self.gamma) * self.sigmoid_cross_entropy_with_logits(input, target)
        loss = loss.squeeze(-1)
        assert weights.dim() == loss.dim()
        return loss * weights


def compute_points_obj_cls_loss_hard_topk(end_points, topk):
    """Seed objectness loss (losses.py:161-223): the `topk` seeds closest (size-normalised) to each valid
    ground-truth centre among the seeds that belong to that object are positives."""
    mask = end_points["box_label_mask"]                                  # (B, G)
    seed_inds = end_points["seed_inds"].long()                           # (B, K)
    seed_xyz = end_points["seed_xyz"]                                    # (B, K, 3)
    logits = end_points["seeds_obj_cls_logits"]                          # (B, 1, K)
    gt_center = end_points["center_label"][:, :, :3]
    gt_size = end_points["size_gts"][:, :, :3]
    B, K, G = gt_center.shape[0], seed_xyz.shape[1], gt_center.shape[1]
    if _fused(logits):
        return _SeedObjectness.apply(logits, seed_xyz, end_points["seed_inds"], end_points["point_instance_label"],
                                     gt_center, gt_size, mask, topk)

    seed_obj = torch.gather(end_points["point_instance_label"], 1, seed_inds)  # (B, K), < 0 = background
    owner = torch.where(seed_obj < 0, torch.full_like(seed_obj, G - 1), seed_obj)
    owned = owner[:, None, :] == torch.arange(G, device=owner.device)[None, :, None]  # (B, G, K)
    delta = (seed_xyz[:, None, :, :] - gt_center[:, :, None, :]) / (gt_size[:, :, None, :] + 1e-6)
    d = torch.sqrt(torch.sum(delta ** 2, dim=-1) + 1e-6)                # (B, G, K)
    d = torch.where(owned, d, torch.full_like(d, 100.0))
    near = torch.topk(d, topk, largest=False)[1]                         # (B, G, topk)
    near = torch.where(mask[:, :, None] > 0, near, torch.full_like(near, K))  # padded boxes -> spare column
    label = torch.zeros((B, K + 1), dtype=torch.long, device=seed_xyz.device)
    label.scatter_(1, near.reshape(B, -1), 1)
    label = label[:, :K] * (seed_obj >= 0).long()

    weights = torch.full((B, K), 1.0 / max(K, 1), device=seed_xyz.device)
    loss = SigmoidFocalClassificationLoss()(logits.reshape(B, K, 1), label.unsqueeze(-1).float(), weights)
    return loss.sum() / B


# ------------------------------------------------------------------------------------------ fused terms
_BACKEND = "hip"


def set_backend(name):
    """'hip' (default): CUDA tensors take the fused kernels of include/butd_criterion.h; 'torch': the dense
    torch expressions below everywhere (the A/B leg; CPU tensors always take them -- they are what the
    CPU tests check against the reference's vectors)."""
    global _BACKEND
    assert name in ("hip", "torch")
    _BACKEND = name


def _fused(t):
    return _BACKEND == "hip" and t.is_cuda


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _f32c(t):
    return t.detach().contiguous().float()


def _pad_last(t, width):
    return t if t.shape[-1] >= width else torch.nn.functional.pad(t, (0, width - t.shape[-1]))


class _BoxLoss(torch.autograd.Function):
    """butd_box_loss: (P,2) sums of the L1 and the 1-GIoU terms over the matched pairs."""

    @staticmethod
    def forward(ctx, pred_boxes, tgt_boxes, match):
        P, B, Q, _ = pred_boxes.shape
        G = tgt_boxes.shape[1]
        pred, tgt, match = _f32c(pred_boxes), _f32c(tgt_boxes), match.contiguous()
        sums = torch.empty((P, 2), device=pred.device)
        grad = torch.empty((P, B, G, 12), device=pred.device)
        lib = _hiplib.load()
        with torch.cuda.device(pred.device):
            _hiplib.check(lib.butd_box_loss(P, B, Q, G, pred.data_ptr(), tgt.data_ptr(), match.data_ptr(),
                                            sums.data_ptr(), grad.data_ptr(), _stream(pred)), "butd_box_loss")
        ctx.save_for_backward(match, grad)
        ctx.dims = (P, B, Q, G)
        return sums

    @staticmethod
    def backward(ctx, g):
        match, grad = ctx.saved_tensors
        P, B, Q, G = ctx.dims
        w = _f32c(g)
        out = torch.empty((P, B, Q, 6), device=grad.device)
        lib = _hiplib.load()
        with torch.cuda.device(grad.device):
            _hiplib.check(lib.butd_box_loss_bwd(P, B, Q, G, match.data_ptr(), grad.data_ptr(), w.data_ptr(),
                                                out.data_ptr(), _stream(grad)), "butd_box_loss_bwd")
        return out, None, None


class _SoftTokenCE(torch.autograd.Function):
    """butd_soft_token_ce: (P,) weighted soft-token cross entropy summed over scenes and queries."""

    @staticmethod
    def forward(ctx, logits, match, positive_map, eos_coef):
        P, B, Q, C = logits.shape
        x, pm, match = _f32c(logits), _f32c(_pad_last(positive_map, C)), match.contiguous()
        G = pm.shape[1]
        rows = torch.empty((P, B, Q), device=x.device)
        dx = torch.empty_like(x)
        lib = _hiplib.load()
        with torch.cuda.device(x.device):
            _hiplib.check(lib.butd_soft_token_ce(P, B, Q, G, C, x.data_ptr(), match.data_ptr(), pm.data_ptr(),
                                                 pm.shape[-1], float(eos_coef), rows.data_ptr(), dx.data_ptr(),
                                                 _stream(x)), "butd_soft_token_ce")
        ctx.save_for_backward(dx)
        return rows.sum((1, 2))

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return dx * g[:, None, None, None], None, None, None


class _ContrastiveAlign(torch.autograd.Function):
    """butd_contrastive_rows + _cols: (P,) (box-to-token + token-to-box) / 2."""

    @staticmethod
    def forward(ctx, logits, match, positive_map, last, eos_coef):
        P, B, Q, L = logits.shape
        x, pm, match = _f32c(logits), _f32c(_pad_last(positive_map, L)), match.contiguous()
        last = last.to(torch.int32).contiguous()
        G = pm.shape[1]
        rows = torch.empty((P, B, Q), device=x.device)
        cols = torch.empty((P, B, L), device=x.device)
        owner = torch.empty((P, B, Q), dtype=torch.int32, device=x.device)
        dx = torch.empty_like(x)
        lib = _hiplib.load()
        with torch.cuda.device(x.device):
            _hiplib.check(lib.butd_contrastive_rows(P, B, Q, G, L, x.data_ptr(), match.data_ptr(), pm.data_ptr(),
                                                    pm.shape[-1], last.data_ptr(), float(eos_coef),
                                                    rows.data_ptr(), dx.data_ptr(), owner.data_ptr(), _stream(x)),
                          "butd_contrastive_rows")
            _hiplib.check(lib.butd_contrastive_cols(P, B, Q, G, L, x.data_ptr(), owner.data_ptr(), pm.data_ptr(),
                                                    pm.shape[-1], last.data_ptr(), float(eos_coef),
                                                    cols.data_ptr(), dx.data_ptr(), _stream(x)),
                          "butd_contrastive_cols")
        ctx.save_for_backward(dx)
        return rows.sum((1, 2)) + cols.sum((1, 2))

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return dx * g[:, None, None, None], None, None, None, None


class _SeedObjectness(torch.autograd.Function):
    """butd_seed_objectness: scalar loss of compute_points_obj_cls_loss_hard_topk."""

    @staticmethod
    def forward(ctx, logits, seed_xyz, seed_inds, pil, gt_center, gt_size, mask, topk):
        B, K = seed_xyz.shape[:2]
        G, N = gt_center.shape[1], pil.shape[1]
        x = _f32c(logits).reshape(B, K)
        xyz, gc, gs, bm = _f32c(seed_xyz), _f32c(gt_center), _f32c(gt_size), _f32c(mask)
        inds, pil = seed_inds.to(torch.int32).contiguous(), pil.to(torch.int64).contiguous()
        label = torch.empty((B, K), dtype=torch.uint8, device=x.device)
        elem, dx = torch.empty_like(x), torch.empty_like(x)
        lib = _hiplib.load()
        with torch.cuda.device(x.device):
            _hiplib.check(lib.butd_seed_objectness(B, K, G, N, int(topk), xyz.data_ptr(), inds.data_ptr(),
                                                   pil.data_ptr(), gc.data_ptr(), gs.data_ptr(), bm.data_ptr(),
                                                   x.data_ptr(), label.data_ptr(), elem.data_ptr(), dx.data_ptr(),
                                                   _stream(x)), "butd_seed_objectness")
        ctx.save_for_backward(dx)
        ctx.shape, ctx.B = logits.shape, B
        return elem.sum() / B

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return (dx * (g / ctx.B)).reshape(ctx.shape), None, None, None, None, None, None, None


def _fused_cost(pred_boxes, tgt_boxes, valid, class_cost, w_bbox, w_class, w_giou):
    """butd_match_cost: pred_boxes (P,B,Q,6), class_cost (P,B,G,Q) or None -> cost (P,B,G,Q)."""
    P, B, Q, _ = pred_boxes.shape
    G = tgt_boxes.shape[1]
    pred, tgt = _f32c(pred_boxes), _f32c(tgt_boxes)
    ok = valid.to(torch.uint8).contiguous()
    cc = _f32c(class_cost) if class_cost is not None else None
    cost = torch.empty((P, B, G, Q), device=pred.device)
    lib = _hiplib.load()
    with torch.cuda.device(pred.device):
        _hiplib.check(lib.butd_match_cost(P, B, Q, G, pred.data_ptr(), tgt.data_ptr(), ok.data_ptr(),
                                          cc.data_ptr() if cc is not None else None, float(w_bbox),
                                          float(w_class), float(w_giou), cost.data_ptr(), _stream(pred)),
                      "butd_match_cost")
    return cost


# ------------------------------------------------------------------------------------------ matcher
def hungarian_match(cost, valid):
    """cost (..., G, Q) fp32 CUDA, valid (..., G) bool/uint8 -> match (..., G) int32 (-1 = not a target)
    and status (...) int32 (include/butd_lsap.h).  Asynchronous on the current stream."""
    if not cost.is_cuda:
        raise RuntimeError("hungarian_match: CPU not supported (the assignment runs in butd_hungarian_match)")
    lead, (G, Q) = cost.shape[:-2], cost.shape[-2:]
    cost = cost.detach().contiguous().float()
    valid_u8 = valid.expand(*lead, G).to(torch.uint8).contiguous()
    count = 1
    for s in lead:
        count *= s
    match = torch.empty((*lead, G), dtype=torch.int32, device=cost.device)
    status = torch.empty(lead, dtype=torch.int32, device=cost.device)
    lib = _hiplib.load()
    with torch.cuda.device(cost.device):
        err = lib.butd_hungarian_match(count, Q, G, cost.data_ptr(), valid_u8.data_ptr(), match.data_ptr(),
                                       status.data_ptr(), torch.cuda.current_stream(cost.device).cuda_stream)
    _hiplib.check(err, "butd_hungarian_match")
    return match, status


class HungarianMatcher(nn.Module):
    """losses.py:226-331.  `match_dense` is the synchronisation-free form; `forward` keeps the reference's
    (outputs, list-of-dict targets) -> list of (index_i, index_j) contract."""

    def __init__(self, cost_class=1, cost_bbox=5, cost_giou=2, soft_token=False):
        super().__init__()
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou
        self.soft_token = soft_token

    @torch.no_grad()
    def cost(self, pred_logits, pred_boxes, tgt_boxes, positive_map, labels=None, valid=None):
        """pred_logits (..., B, Q, C), pred_boxes (..., B, Q, 6), tgt_boxes (B, G, 6), positive_map
        (B, G, C') -> C (..., B, G, Q): losses.py:285-312 for every target slot."""
        prob = pred_logits.softmax(-1)
        if _fused(pred_logits) and pred_logits.dim() == 4:
            if self.soft_token:
                pm = positive_map[..., :prob.shape[-1]] if positive_map.shape[-1] != prob.shape[-1] else positive_map
                affinity = torch.matmul(pm, prob.transpose(-1, -2))                          # = -cost_class
            else:
                idx = labels.long()[..., :, None].expand(*prob.shape[:-2], labels.shape[-1], prob.shape[-2])
                affinity = torch.gather(prob.transpose(-1, -2), -2, idx)
            if valid is None:   # rows of slots that are not targets are skipped (written as zeros)
                valid = torch.ones(tgt_boxes.shape[:2], dtype=torch.uint8, device=tgt_boxes.device)
            return _fused_cost(pred_boxes, tgt_boxes, valid, affinity, self.cost_bbox, -self.cost_class,
                               self.cost_giou)
        if self.soft_token:
            pm = positive_map[..., :prob.shape[-1]] if positive_map.shape[-1] != prob.shape[-1] else positive_map
            cost_class = -torch.matmul(pm, prob.transpose(-1, -2))                        # (..., B, G, Q)
        else:
            idx = labels.long()[..., :, None].expand(*prob.shape[:-2], labels.shape[-1], prob.shape[-2])
            cost_class = -torch.gather(prob.transpose(-1, -2), -2, idx)
        cost_bbox = (tgt_boxes[..., :, None, :] - pred_boxes[..., None, :, :]).abs().sum(-1)
        cost_giou = -_giou_broadcast(box_cxcyczwhd_to_xyzxyz(tgt_boxes)[..., :, None, :],
                                     box_cxcyczwhd_to_xyzxyz(pred_boxes)[..., None, :, :])
        return self.cost_bbox * cost_bbox + self.cost_class * cost_class + self.cost_giou * cost_giou

    @torch.no_grad()
    def match_dense(self, pred_logits, pred_boxes, tgt_boxes, positive_map, valid, labels=None):
        """-> match (..., B, G).  ``self.last_status`` keeps the solver's per-problem status tensor (1 where scipy
        would have raised: NaN / -inf costs, no feasible assignment) for callers that choose to look."""
        match, self.last_status = hungarian_match(
            self.cost(pred_logits, pred_boxes, tgt_boxes, positive_map, labels, valid), valid)
        return match

    @torch.no_grad()
    def forward(self, outputs, targets):
        tgt_boxes, pm, labels, valid = _pad_targets(targets, outputs["pred_logits"].shape[-1])
        match = self.match_dense(outputs["pred_logits"], outputs["pred_boxes"], tgt_boxes, pm, valid, labels)
        return _match_to_indices(match, valid)


def _pad_targets(targets, n_class):
    """list of {'labels' (n,), 'boxes' (n,6), 'positive_map' (n,C)} -> padded (B,G,.) tensors + mask."""
    G = max(1, max(len(t["boxes"]) for t in targets))
    dev = targets[0]["boxes"].device
    B = len(targets)
    boxes = torch.zeros((B, G, 6), device=dev)
    boxes[..., 3:] = 1.0
    width = targets[0]["positive_map"].shape[-1] if "positive_map" in targets[0] else n_class
    pm = torch.zeros((B, G, width), device=dev)
    labels = torch.zeros((B, G), dtype=torch.long, device=dev)
    valid = torch.zeros((B, G), dtype=torch.bool, device=dev)
    for b, t in enumerate(targets):
        n = len(t["boxes"])
        boxes[b, :n] = t["boxes"]
        if "positive_map" in t:
            pm[b, :n] = t["positive_map"]
        labels[b, :n] = t["labels"]
        valid[b, :n] = True
    return boxes, pm, labels, valid


def _match_to_indices(match, valid):
    """match (B, G) -> the reference's list of (queries ascending, targets) int64 CPU pairs."""
    out = []
    match, valid = match.cpu(), valid.cpu()
    for b in range(match.shape[0]):
        slots = torch.nonzero(valid[b]).flatten()
        q = match[b, slots].long()
        order = torch.argsort(q)
        compact = torch.arange(len(slots))
        out.append((q[order], compact[order]))
    return out


# ------------------------------------------------------------------------------------------ criterion
class SetCriterion(nn.Module):
    """losses.py:334-543 on dense targets.  `losses` is the reference's list: any of 'boxes', 'labels',
    'contrastive_align'."""

    def __init__(self, matcher, losses={}, eos_coef=0.1, temperature=0.07):
        super().__init__()
        self.matcher, self.eos_coef, self.losses, self.temperature = matcher, eos_coef, losses, temperature

    # -- helpers: rows of a (P, B, Q+1, .) tensor addressed by target slot; invalid slots hit the spare row Q
    def _scatter_rows(self, base, match, valid, rows):
        """base (P,B,Q,W) <- rows (B,G,W) at query match[p,b,g] for valid slots."""
        P, B, Q, W = base.shape
        idx = torch.where(valid, match.long(), torch.full_like(match, Q, dtype=torch.long))  # (P,B,G)
        padded = torch.cat([base, base.new_zeros(P, B, 1, W)], dim=2)
        src = rows[None].expand(P, -1, -1, -1)
        padded = padded.scatter(2, idx[..., None].expand(-1, -1, -1, W), src.to(base.dtype))
        return padded[:, :, :Q]

    def _matched_mask(self, match, valid, Q):
        P, B, G = match.shape
        idx = torch.where(valid, match.long(), torch.full_like(match, Q, dtype=torch.long))
        m = torch.zeros((P, B, Q + 1), dtype=torch.bool, device=match.device)
        return m.scatter(2, idx, torch.ones_like(idx, dtype=torch.bool))[:, :, :Q]

    def loss_labels_st(self, out, tgt, match, num_boxes):
        if _fused(out["pred_logits"]):
            ce = _SoftTokenCE.apply(out["pred_logits"], match, tgt["positive_map"], self.eos_coef)
            return {"loss_ce": ce / num_boxes}
        logits = out["pred_logits"].log_softmax(-1)                               # (P,B,Q,C)
        P, B, Q, C = logits.shape
        valid = tgt["valid"][None].expand(P, -1, -1)
        target_sim = torch.zeros_like(logits)
        target_sim[..., -1] = 1
        target_sim = self._scatter_rows(target_sim, match, valid, tgt["positive_map"][..., :C])
        entropy = torch.log(target_sim + 1e-6) * target_sim
        loss_ce = (entropy - logits * target_sim).sum(-1)                         # (P,B,Q)
        matched = self._matched_mask(match, valid, Q)
        weight = torch.where(matched, torch.ones_like(loss_ce), torch.full_like(loss_ce, self.eos_coef))
        return {"loss_ce": (loss_ce * weight).sum((1, 2)) / num_boxes}

    def loss_boxes(self, out, tgt, match, num_boxes):
        if _fused(out["pred_boxes"]):
            sums = _BoxLoss.apply(out["pred_boxes"], tgt["boxes"], match)
            return {"loss_bbox": sums[:, 0] / num_boxes, "loss_giou": sums[:, 1] / num_boxes}
        P = match.shape[0]
        valid = tgt["valid"][None].expand(P, -1, -1)
        idx = match.long().clamp(min=0)
        src = torch.gather(out["pred_boxes"], 2, idx[..., None].expand(-1, -1, -1, 6))   # (P,B,G,6)
        box = tgt["boxes"][None].expand(P, -1, -1, -1)
        l1 = (src[..., :3] - box[..., :3]).abs().sum(-1) + 0.2 * (src[..., 3:] - box[..., 3:]).abs().sum(-1)
        zero = torch.zeros_like(l1)
        giou = _giou_broadcast(box_cxcyczwhd_to_xyzxyz(src), box_cxcyczwhd_to_xyzxyz(box))
        return {"loss_bbox": torch.where(valid, l1, zero).sum((1, 2)) / num_boxes,
                "loss_giou": torch.where(valid, 1 - giou, zero).sum((1, 2)) / num_boxes}

    def loss_contrastive_align(self, out, tgt, match, num_boxes):
        tok, que = out["proj_tokens"], out["proj_queries"]                        # (B,L,D), (P,B,Q,D)
        logits = torch.matmul(que, tok.transpose(-1, -2)) / self.temperature      # (P,B,Q,L)
        P, B, Q, L = logits.shape
        valid = tgt["valid"][None].expand(P, -1, -1)
        # 'not mentioned': the last two real tokens (python indexing: -1 wraps to the last column)
        inds = out["tokenized"]["attention_mask"].to(logits.device).sum(1) - 1    # (B,)
        if _fused(logits):
            align = _ContrastiveAlign.apply(logits, match, tgt["positive_map"], inds, self.eos_coef)
            return {"loss_contrastive_align": align / num_boxes}
        cols = torch.arange(L, device=logits.device)[None, :]
        base = ((cols == inds[:, None]) | (cols == torch.remainder(inds[:, None] - 1, L))).to(logits.dtype) * 0.5
        pmap = base[None, :, None, :].expand(P, -1, Q, -1)
        pm_rows = tgt["positive_map"][..., :L]
        if pm_rows.shape[-1] < L:
            pm_rows = torch.nn.functional.pad(pm_rows, (0, L - pm_rows.shape[-1]))
        positive = self._scatter_rows(pmap.contiguous(), match, valid, pm_rows) > 0
        matched = self._matched_mask(match, valid, Q)
        mask = torch.where(matched, torch.ones((), device=logits.device),
                           torch.full((), self.eos_coef, device=logits.device))   # (P,B,Q)
        tmask = torch.where(cols == inds[:, None], torch.ones((), device=logits.device),
                            torch.full((), self.eos_coef, device=logits.device))  # (B,L)

        positive_logits = -logits.masked_fill(~positive, 0)

        def side(dim, weights):
            has_pos = positive.any(dim)
            pos_term = positive_logits.sum(dim)
            neg_term = logits.logsumexp(dim)
            nb_pos = positive.sum(dim) + 1e-6
            entropy = -torch.log(nb_pos + 1e-6) / nb_pos
            per = (entropy + pos_term / nb_pos + neg_term).masked_fill(~has_pos, 0)
            return (per * weights).sum((1, 2))

        box_to_token = side(3, mask)
        token_to_box = side(2, tmask[None])
        return {"loss_contrastive_align": (box_to_token + token_to_box) / 2 / num_boxes}

    def get_loss(self, loss, out, tgt, match, num_boxes):
        table = {"labels": self.loss_labels_st, "boxes": self.loss_boxes,
                 "contrastive_align": self.loss_contrastive_align}
        assert loss in table, f"do you really want to compute {loss} loss?"
        return table[loss](out, tgt, match, num_boxes)

    def num_boxes(self, valid):
        """losses.py:527-534 as a device scalar: matched pairs of this rank, averaged over ranks, >= 1."""
        n = valid.sum().to(torch.float32).reshape(1)
        world = 1
        if is_dist_avail_and_initialized():
            dist.all_reduce(n)
            world = dist.get_world_size()
        return torch.clamp(n / world, min=1)

    def dense_forward(self, out, tgt, match=None):
        """out: 'pred_logits' (P,B,Q,C), 'pred_boxes' (P,B,Q,6) [, 'proj_queries' (P,B,Q,D), 'proj_tokens'
        (B,L,D), 'tokenized'];  tgt: 'boxes' (B,G,6), 'positive_map' (B,G,C), 'labels' (B,G), 'valid' (B,G)
        bool.  Returns ({name: (P,) tensor}, match (P,B,G)).  `match` may be supplied (tests)."""
        if match is None:
            match = self.matcher.match_dense(out["pred_logits"], out["pred_boxes"], tgt["boxes"],
                                             tgt["positive_map"], tgt["valid"], tgt.get("labels"))
        num_boxes = tgt["num_boxes"] if tgt.get("num_boxes") is not None else self.num_boxes(tgt["valid"])
        losses = {}
        for name in self.losses:
            losses.update(self.get_loss(name, out, tgt, match, num_boxes))
        return losses, match

    def forward(self, outputs, targets):
        """The reference's call: one prefix, list-of-dict targets -> (losses, indices)."""
        boxes, pm, labels, valid = _pad_targets(targets, outputs["pred_logits"].shape[-1])
        out = {k: (v[None] if k in ("pred_logits", "pred_boxes", "proj_queries") else v) for k, v in outputs.items()}
        losses, match = self.dense_forward(out, {"boxes": boxes, "positive_map": pm, "labels": labels,
                                                 "valid": valid})
        return {k: v[0] for k, v in losses.items()}, _match_to_indices(match[0], valid)


class _CombineLosses(torch.autograd.Function):
    """losses.py:592-617: sums over the prefixes + the weighting (+ the NaN of a failed assignment) as one launch
    forward, one backward (``butd_loss_combine``); the stock expression is ~20 scalar launches each way."""

    @staticmethod
    def forward(ctx, ce, bbox, giou, align, generation, status, w_gen, w_sum, w_bbox):
        lib = _hiplib.load()
        P = bbox.numel()
        out = torch.empty(5, device=bbox.device)
        terms = [None if t is None else _f32c(t) for t in (ce, bbox, giou, align, generation)]
        st = None if status is None else status.detach().contiguous().view(-1).to(torch.int32)
        ptr = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(bbox.device):
            _hiplib.check(lib.butd_loss_combine(P, ptr(terms[0]), ptr(terms[1]), ptr(terms[2]), ptr(terms[3]),
                                                ptr(terms[4]), ptr(st), 0 if st is None else st.numel(), w_gen, w_sum,
                                                w_bbox, out.data_ptr(), _stream(bbox)), "butd_loss_combine")
        ctx.cfg = (P, w_gen, w_sum, w_bbox, [None if t is None else t.shape for t in (ce, bbox, giou, align, generation)])
        ctx.status = st         # (not differentiable; the backward zeroes the gradients of an invalid match)
        parts = out.unbind(0)
        ctx.mark_non_differentiable(*parts[1:])
        return parts

    @staticmethod
    def backward(ctx, g, *_unused):
        P, w_gen, w_sum, w_bbox, shapes = ctx.cfg
        lib = _hiplib.load()
        g = g.contiguous()
        grads = [None if sh is None else torch.empty(sh, device=g.device) for sh in shapes]
        ptr = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(g.device):
            st = ctx.status
            _hiplib.check(lib.butd_loss_combine_bwd(P, g.data_ptr(), ptr(st), 0 if st is None else st.numel(), w_gen,
                                                    w_sum, w_bbox, ptr(grads[0]), ptr(grads[1]), ptr(grads[2]),
                                                    ptr(grads[3]), ptr(grads[4]), _stream(g)),
                          "butd_loss_combine_bwd")
        return (*grads, None, None, None, None)


class _HungarianTail(torch.autograd.Function):
    """Everything of compute_hungarian_loss behind the assignment as ONE autograd node (round 5): the term kernels of
    include/butd_criterion.h back to back, then ``butd_criterion_reduce`` (row sums, the divisions by num_boxes, the sums
    over the prefixes, the weighting, the NaN of a failed assignment: one launch); backward: ``butd_criterion_scale``
    (the saved derivatives times g w / num_boxes, three tensors in one launch) + ``butd_box_loss_bwd``.  As separate
    autograd nodes glued by torch expressions the same arithmetic was ~65 graph nodes of 4.6 us each.
    Returns (loss, [ce, bbox, giou, align, generation] summed over the prefixes, per_prefix (P,4) = the four terms)."""

    @staticmethod
    def forward(ctx, pred_logits, pred_boxes, align_logits, seed_logits, cfg):
        lib = _hiplib.load()
        dev = pred_boxes.device
        P, B, Q, _ = pred_boxes.shape
        match, tgt_boxes, pmap, num_boxes, status = cfg["match"], cfg["tgt_boxes"], cfg["positive_map"], cfg["num_boxes"], cfg["status"]
        G = tgt_boxes.shape[1]
        eos = float(cfg["eos_coef"])
        match = match.contiguous()
        ptr = lambda t: None if t is None else t.data_ptr()
        st = None if status is None else status.detach().contiguous().view(-1).to(torch.int32)
        nb = _f32c(num_boxes).reshape(-1)
        with torch.cuda.device(dev):
            stream = _stream(pred_boxes)
            pred, tgt = _f32c(pred_boxes), _f32c(tgt_boxes)
            sums = torch.empty((P, 2), device=dev)
            box_grad = torch.empty((P, B, G, 12), device=dev)
            _hiplib.check(lib.butd_box_loss(P, B, Q, G, pred.data_ptr(), tgt.data_ptr(), match.data_ptr(), sums.data_ptr(),
                                            box_grad.data_ptr(), stream), "butd_box_loss")
            ce_rows = dx_ce = None
            if pred_logits is not None:
                C = pred_logits.shape[-1]
                x, pm = _f32c(pred_logits), _f32c(_pad_last(pmap, C))
                ce_rows, dx_ce = torch.empty((P, B, Q), device=dev), torch.empty_like(x)
                _hiplib.check(lib.butd_soft_token_ce(P, B, Q, G, C, x.data_ptr(), match.data_ptr(), pm.data_ptr(),
                                                     pm.shape[-1], eos, ce_rows.data_ptr(), dx_ce.data_ptr(), stream),
                              "butd_soft_token_ce")
            rows = cols = dx_al = None
            if align_logits is not None:
                L_ = align_logits.shape[-1]
                x, pm = _f32c(align_logits), _f32c(_pad_last(pmap, L_))
                last = cfg["last"].to(torch.int32).contiguous()
                rows, cols = torch.empty((P, B, Q), device=dev), torch.empty((P, B, L_), device=dev)
                owner = torch.empty((P, B, Q), dtype=torch.int32, device=dev)
                dx_al = torch.empty_like(x)
                _hiplib.check(lib.butd_contrastive_rows(P, B, Q, G, L_, x.data_ptr(), match.data_ptr(), pm.data_ptr(),
                                                        pm.shape[-1], last.data_ptr(), eos, rows.data_ptr(),
                                                        dx_al.data_ptr(), owner.data_ptr(), stream), "butd_contrastive_rows")
                _hiplib.check(lib.butd_contrastive_cols(P, B, Q, G, L_, x.data_ptr(), owner.data_ptr(), pm.data_ptr(),
                                                        pm.shape[-1], last.data_ptr(), eos, cols.data_ptr(),
                                                        dx_al.data_ptr(), stream), "butd_contrastive_cols")
            elem = dx_gen = None
            gen_div = 1.0
            if seed_logits is not None:
                g = cfg["generation"]
                K, N = g["seed_xyz"].shape[1], g["pil"].shape[1]
                x = _f32c(seed_logits).reshape(B, K)
                xyz, gc, gs, bm = _f32c(g["seed_xyz"]), _f32c(g["gt_center"]), _f32c(g["gt_size"]), _f32c(g["mask"])
                inds, pil = g["seed_inds"].to(torch.int32).contiguous(), g["pil"].to(torch.int64).contiguous()
                label = torch.empty((B, K), dtype=torch.uint8, device=dev)
                elem, dx_gen = torch.empty_like(x), torch.empty_like(x)
                _hiplib.check(lib.butd_seed_objectness(B, K, gc.shape[1], N, int(g["topk"]), xyz.data_ptr(), inds.data_ptr(),
                                                       pil.data_ptr(), gc.data_ptr(), gs.data_ptr(), bm.data_ptr(),
                                                       x.data_ptr(), label.data_ptr(), elem.data_ptr(), dx_gen.data_ptr(),
                                                       stream), "butd_seed_objectness")
                gen_div = float(B)
            buf = torch.empty(6 + 4 * P, device=dev)          # (the outputs are views of it, as _CombineLosses' were)
            out6, per_prefix = buf[:6], buf[6:].view(P, 4)
            w_gen, w_sum, w_bbox = cfg["weights"]
            _hiplib.check(lib.butd_criterion_reduce(
                P, ptr(ce_rows), 0 if ce_rows is None else B * Q, sums.data_ptr(), ptr(rows), 0 if rows is None else B * Q,
                ptr(cols), 0 if cols is None else cols.shape[1] * cols.shape[2], ptr(elem), 0 if elem is None else elem.numel(),
                gen_div, nb.data_ptr(), ptr(st), 0 if st is None else st.numel(), w_gen, w_sum, w_bbox,
                per_prefix.data_ptr(), out6.data_ptr(), stream), "butd_criterion_reduce")
        ctx.save_for_backward(match, box_grad, dx_ce, dx_al, dx_gen, nb, st)
        ctx.cfg = (P, B, Q, G, w_gen, w_sum, w_bbox, gen_div,
                   None if pred_logits is None else pred_logits.shape, None if align_logits is None else align_logits.shape,
                   None if seed_logits is None else seed_logits.shape)
        sums5 = buf[1:6]
        ctx.mark_non_differentiable(sums5, per_prefix)
        return buf[0], sums5, per_prefix

    @staticmethod
    def backward(ctx, g, *_unused):
        match, box_grad, dx_ce, dx_al, dx_gen, nb, st = ctx.saved_tensors
        P, B, Q, G, w_gen, w_sum, w_bbox, gen_div, sh_ce, sh_al, sh_gen = ctx.cfg
        lib = _hiplib.load()
        dev = box_grad.device
        g = _f32c(g).reshape(1)
        ptr = lambda t: None if t is None else t.data_ptr()
        d_ce = None if dx_ce is None else torch.empty_like(dx_ce)
        d_al = None if dx_al is None else torch.empty_like(dx_al)
        d_gen = None if dx_gen is None else torch.empty_like(dx_gen)
        box_w = torch.empty((P, 2), device=dev)
        d_boxes = torch.empty((P, B, Q, 6), device=dev)
        with torch.cuda.device(dev):
            stream = _stream(box_grad)
            _hiplib.check(lib.butd_criterion_scale(
                P, g.data_ptr(), nb.data_ptr(), ptr(st), 0 if st is None else st.numel(), w_gen, w_sum, w_bbox, gen_div,
                ptr(dx_ce), ptr(d_ce), 0 if dx_ce is None else dx_ce.numel(), ptr(dx_al), ptr(d_al),
                0 if dx_al is None else dx_al.numel(), ptr(dx_gen), ptr(d_gen), 0 if dx_gen is None else dx_gen.numel(),
                box_w.data_ptr(), stream), "butd_criterion_scale")
            _hiplib.check(lib.butd_box_loss_bwd(P, B, Q, G, match.data_ptr(), box_grad.data_ptr(), box_w.data_ptr(),
                                                d_boxes.data_ptr(), stream), "butd_box_loss_bwd")
        return (d_ce if d_ce is None else d_ce.view(sh_ce), d_boxes, d_al if d_al is None else d_al.view(sh_al),
                d_gen if d_gen is None else d_gen.view(sh_gen), None)


def hungarian_prefixes(num_decoder_layers):
    return ["proposal_", "last_"] + [f"{i}head_" for i in range(num_decoder_layers - 1)]


_TAIL = [True]      # (tests: the single-node tail of compute_hungarian_loss vs the term-by-term autograd graph)


def _hungarian_tail(end_points, prefixes, num_decoder_layers, crit, out, tgt, match, topk):
    """compute_hungarian_loss behind the stacking of the prefixes, CUDA tensors, fused backend: the assignment, then
    ``_HungarianTail`` (one autograd node).  Writes the same keys into ``end_points`` as the term-by-term path."""
    if match is None:
        match = crit.matcher.match_dense(out["pred_logits"], out["pred_boxes"], tgt["boxes"], tgt["positive_map"],
                                         tgt["valid"], tgt.get("labels"))
    status = getattr(crit.matcher, "last_status", None)
    num_boxes = tgt["num_boxes"] if tgt.get("num_boxes") is not None else crit.num_boxes(tgt["valid"])
    cfg = {"match": match, "tgt_boxes": tgt["boxes"], "positive_map": tgt["positive_map"], "num_boxes": num_boxes,
           "status": status, "eos_coef": crit.eos_coef, "weights": (8.0, 1.0 / (num_decoder_layers + 1), 5.0)}
    align_logits = None
    if "contrastive_align" in crit.losses and "proj_tokens" in end_points:
        align_logits = torch.matmul(out["proj_queries"], out["proj_tokens"].transpose(-1, -2)) / crit.temperature
        # 'not mentioned': the last real token (python indexing: -1 wraps to the last column)
        cfg["last"] = out["tokenized"]["attention_mask"].to(align_logits.device).sum(1) - 1
    seed_logits = None
    if "seeds_obj_cls_logits" in end_points:
        seed_logits = end_points["seeds_obj_cls_logits"]
        cfg["generation"] = {"seed_xyz": end_points["seed_xyz"], "seed_inds": end_points["seed_inds"],
                             "pil": end_points["point_instance_label"], "gt_center": end_points["center_label"][:, :, :3],
                             "gt_size": end_points["size_gts"][:, :, :3], "mask": end_points["box_label_mask"], "topk": topk}
    logits = out["pred_logits"] if "labels" in crit.losses else None
    loss, sums5, per_prefix = _HungarianTail.apply(logits, out["pred_boxes"], align_logits, seed_logits, cfg)
    names = (("loss_ce", 0, logits is not None), ("loss_bbox", 1, True), ("loss_giou", 2, True),
             ("loss_contrastive_align", 3, align_logits is not None))
    for key, col, present in names:
        if present:
            for i, p in enumerate(prefixes):
                end_points[f"{p}_{key}"] = per_prefix[i, col]
    end_points.update({"loss_ce": sums5[0], "loss_bbox": sums5[1], "loss_giou": sums5[2],
                       "query_points_generation_loss": sums5[4], "loss_constrastive_align": sums5[3], "loss": loss,
                       "hungarian_match": match, "hungarian_status": status})
    return loss, end_points


def compute_hungarian_loss(end_points, num_decoder_layers, set_criterion, query_points_obj_topk=5, match=None):
    """losses.py:546-617: same keys written into `end_points`, same weighting; every prefix evaluated in one
    stacked pass."""
    prefixes = hungarian_prefixes(num_decoder_layers)
    valid = end_points["box_label_mask"].bool()
    tgt = {"boxes": torch.cat([end_points["center_label"][:, :, 0:3], end_points["size_gts"]], dim=-1),
           "positive_map": end_points["positive_map"], "labels": end_points["sem_cls_label"], "valid": valid,
           "num_boxes": end_points.get("num_boxes")}   # optional: precomputed rank-averaged count (device)
    stack = lambda key: torch.stack([end_points[f"{p}{key}"] for p in prefixes])
    out = {"pred_logits": stack("sem_cls_scores"),
           "pred_boxes": torch.cat([stack("center"), stack("pred_size")], dim=-1)}
    if "proj_tokens" in end_points:
        out["proj_tokens"] = end_points["proj_tokens"]
        out["proj_queries"] = stack("proj_queries")
        out["tokenized"] = end_points["tokenized"]
    if (_TAIL[0] and _fused(out["pred_boxes"]) and len(prefixes) <= 64 and "boxes" in set_criterion.losses
            and set(set_criterion.losses) <= {"boxes", "labels", "contrastive_align"}):
        return _hungarian_tail(end_points, prefixes, num_decoder_layers, set_criterion, out, tgt, match,
                               query_points_obj_topk)
    losses, match = set_criterion.dense_forward(out, tgt, match)
    for key, per_prefix in losses.items():
        for i, p in enumerate(prefixes):
            end_points[f"{p}_{key}"] = per_prefix[i]
    status = getattr(set_criterion.matcher, "last_status", None)
    if _fused(out["pred_boxes"]) and len(prefixes) <= 64:
        generation = (compute_points_obj_cls_loss_hard_topk(end_points, query_points_obj_topk)
                      if "seeds_obj_cls_logits" in end_points else None)
        loss, loss_ce, loss_bbox, loss_giou, loss_align = _CombineLosses.apply(
            losses.get("loss_ce"), losses["loss_bbox"], losses.get("loss_giou"),
            losses["loss_contrastive_align"] if "proj_tokens" in end_points else None,
            generation, status, 8.0, 1.0 / (num_decoder_layers + 1), 5.0)
        end_points.update({"loss_ce": loss_ce, "loss_bbox": loss_bbox, "loss_giou": loss_giou,
                           "query_points_generation_loss": generation if generation is not None else loss_ce * 0,
                           "loss_constrastive_align": loss_align, "loss": loss, "hungarian_match": match,
                           "hungarian_status": status})
        return loss, end_points
    zero = out["pred_boxes"].new_zeros(())
    loss_ce = losses["loss_ce"].sum() if "loss_ce" in losses else zero
    loss_bbox = losses["loss_bbox"].sum()
    loss_giou = losses["loss_giou"].sum() if "loss_giou" in losses else zero
    loss_align = losses["loss_contrastive_align"].sum() if "proj_tokens" in end_points else zero
    if "seeds_obj_cls_logits" in end_points:
        generation = compute_points_obj_cls_loss_hard_topk(end_points, query_points_obj_topk)
    else:
        generation = zero
    loss = 8 * generation + 1.0 / (num_decoder_layers + 1) * (loss_ce + 5 * loss_bbox + loss_giou + loss_align)
    if status is not None:
        # where scipy's linear_sum_assignment would have raised (NaN / -inf costs: matcher.py:105) the device solver
        # sets a status word instead; inside a graph replay nothing can raise, so the failure is made visible the
        # way the caller does look: the loss of that step is NaN (no host sync)
        loss = torch.where(status.ne(0).any(), torch.full_like(loss, float("nan")), loss)
    end_points["loss_ce"] = loss_ce
    end_points["loss_bbox"] = loss_bbox
    end_points["loss_giou"] = loss_giou
    end_points["query_points_generation_loss"] = generation
    end_points["loss_constrastive_align"] = loss_align
    end_points["loss"] = loss
    end_points["hungarian_match"] = match
    end_points["hungarian_status"] = status
    return loss, end_points
