"""autograd Function API of the reference's ``pointnet2_utils`` on top of the gfx950 operator library.

Names, argument order and autograd contract follow pointnet2/pointnet2_utils.py:
``furthest_point_sample`` (:80), ``gather_operation`` (:117), ``three_nn`` (:149),
``three_interpolate`` (:206), ``grouping_operation`` (:257), ``ball_query`` (:291) -- note the
Python-level order ``ball_query(radius, nsample, xyz, new_xyz)`` vs the extension's
``ball_query(new_xyz, xyz, radius, nsample)`` -- plus ``QueryAndGroup`` (:294-376) and ``GroupAll``
(:379-426).  FPS / ball-query / three_nn outputs are non-differentiable; gather / group /
interpolate back-propagate to ``features`` only.

``_ext`` is the module-level operator backend (``butd_detr_amd.pointnet2_ext``).  It has no CPU
implementation; CPU tensors raise ``RuntimeError("CPU not supported")`` like the reference.
"""
import torch
from torch import nn
from torch.autograd import Function

from . import pointnet2_ext as _ext


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        inds = _ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(inds)
        return inds

    @staticmethod
    def backward(ctx, grad=None):
        return None, None


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.n_source = features.size(2)
        ctx.save_for_backward(idx)
        return _ext.gather_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _ext.gather_points_grad(grad_out.contiguous(), idx, ctx.n_source), None


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        dist2, idx = _ext.three_nn(unknown, known)
        dist = torch.sqrt(dist2)  # the extension returns squared distances (pointnet2_utils.py:142)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, grad_dist=None, grad_idx=None):
        return None, None


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        ctx.m_source = features.size(2)
        ctx.save_for_backward(idx, weight)
        return _ext.three_interpolate(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        grad = _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight, ctx.m_source)
        return grad, None, None


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.n_source = features.size(2)
        ctx.save_for_backward(idx)
        return _ext.group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _ext.group_points_grad(grad_out.contiguous(), idx, ctx.n_source), None


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        inds = _ext.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(inds)
        return inds

    @staticmethod
    def backward(ctx, grad=None):
        return None, None, None, None


furthest_point_sample = FurthestPointSampling.apply
gather_operation = GatherOperation.apply
three_nn = ThreeNN.apply
three_interpolate = ThreeInterpolate.apply
grouping_operation = GroupingOperation.apply
ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """Ball query + grouping: (xyz (B,N,3), new_xyz (B,M,3), features (B,C,N)) ->
    (B, 3+C, M, nsample) with the centre-relative (optionally radius-normalised) xyz channels first
    (pointnet2_utils.py:317-376)."""

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=False,
                 sample_uniformly=False, ret_unique_cnt=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz
        self.normalize_xyz = normalize_xyz
        self.sample_uniformly = sample_uniformly
        self.ret_unique_cnt = ret_unique_cnt
        if ret_unique_cnt:
            assert sample_uniformly

    def _resample_uniformly(self, idx):
        # pointnet2_utils.py:336-345: replace the "pad with first hit" tail by uniform draws
        # from the unique hits (host loop, only used by the votes variant of the modules).
        unique_cnt = torch.zeros(idx.shape[:2])
        for b in range(idx.shape[0]):
            for r in range(idx.shape[1]):
                uniq = torch.unique(idx[b, r, :])
                k = uniq.shape[0]
                unique_cnt[b, r] = k
                draw = torch.randint(0, k, (self.nsample - k,), dtype=torch.long)
                idx[b, r, :] = torch.cat((uniq, uniq[draw]))
        return unique_cnt

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        unique_cnt = self._resample_uniformly(idx) if self.sample_uniformly else None

        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)  # (B,3,M,S)
        grouped_xyz -= new_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped_xyz /= self.radius

        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz
        else:
            grouped = grouping_operation(features, idx)
            new_features = torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped

        out = [new_features]
        if self.ret_grouped_xyz:
            out.append(grouped_xyz)
        if self.ret_unique_cnt:
            out.append(unique_cnt)
        return out[0] if len(out) == 1 else tuple(out)


class GroupAll(nn.Module):
    """Single group containing every point: -> (B, 3+C, 1, N) (pointnet2_utils.py:379-426).
    As in the reference, ``ret_grouped_xyz`` is accepted but forced to False (:390)."""

    def __init__(self, use_xyz=True, ret_grouped_xyz=False):
        super().__init__()
        self.use_xyz = use_xyz
        self.ret_grouped_xyz = False

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            new_features = grouped_xyz
        else:
            grouped = features.unsqueeze(2)
            new_features = torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped
        if self.ret_grouped_xyz:
            return new_features, grouped_xyz
        return new_features
