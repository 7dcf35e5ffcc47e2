"""PointNet++ set-abstraction / feature-propagation modules used by the BUTD-DETR backbone.

Interfaces follow pointnet2/pointnet2_modules.py: ``PointnetSAModuleVotes`` (:164-272) and
``PointnetFPModule`` (:356-416) -- the only two the model instantiates (models/backbone_module.py:32).
Sub-module names (``grouper``, ``mlp_module`` / ``mlp``) match so checkpoints load unchanged.
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import pointnet2_utils
from .pytorch_utils import SharedMLP


class PointnetSAModuleVotes(nn.Module):
    """FPS -> ball-query grouping -> shared MLP -> pool; also returns the sampled indices.

    forward(xyz (B,N,3), features (B,C,N) | None, inds (B,npoint) i32 | None)
        -> new_xyz (B,npoint,3), new_features (B,mlp[-1],npoint), inds (B,npoint) [, unique_cnt]
    """

    def __init__(self, *, mlp, npoint=None, radius=None, nsample=None, bn=True, use_xyz=True,
                 pooling="max", sigma=None, normalize_xyz=False, sample_uniformly=False,
                 ret_unique_cnt=False):
        super().__init__()
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.pooling = pooling
        self.use_xyz = use_xyz
        self.sigma = sigma if sigma is not None else (radius / 2 if radius is not None else None)
        self.normalize_xyz = normalize_xyz
        self.ret_unique_cnt = ret_unique_cnt
        if npoint is not None:
            self.grouper = pointnet2_utils.QueryAndGroup(
                radius, nsample, use_xyz=use_xyz, ret_grouped_xyz=True,
                normalize_xyz=normalize_xyz, sample_uniformly=sample_uniformly,
                ret_unique_cnt=ret_unique_cnt)
        else:
            self.grouper = pointnet2_utils.GroupAll(use_xyz, ret_grouped_xyz=True)
        widths = list(mlp)
        if use_xyz and widths:
            widths[0] += 3
        self.mlp_module = SharedMLP(widths, bn=bn)

    def _pool(self, feats, grouped_xyz):
        if self.pooling == "max":
            return feats.max(dim=3).values
        if self.pooling == "avg":
            return feats.mean(dim=3)
        if self.pooling == "rbf":  # pointnet2_modules.py:263-267
            rbf = torch.exp(-grouped_xyz.pow(2).sum(1) / (self.sigma ** 2) / 2)
            return (feats * rbf.unsqueeze(1)).sum(-1) / float(self.nsample)
        raise ValueError(self.pooling)

    def forward(self, xyz, features=None, inds=None, features_pm=None, feat_offset=0, new_xyz=None, ball_idx=None,
                ball_inv=None):
        """``features_pm`` (optional, not in the reference signature): the same features point-major,
        (B, N, feat_offset + C); lets consecutive levels hand activations over without a transpose.
        After the call ``self.last_features_pm`` holds this level's output point-major (or None).
        ``new_xyz`` / ``ball_idx`` (optional, with ``inds``): the sampled centres and the ball-query neighbour lists as
        ``Pointnet2Backbone.plan`` precomputed them (coordinates only: prefetched for the next batch); ``ball_inv``: the
        inverted neighbour lists (fused_sa.inverse_index) for the feature gradient."""
        self.last_features_pm = None
        if inds is None:
            inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        else:
            assert inds.shape[1] == self.npoint
        if new_xyz is None and self.npoint is not None:
            new_xyz = pointnet2_utils.gather_operation(
                xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
        from . import attention_blocks, fused_sa
        if attention_blocks.get_backend() == "hip" and fused_sa.supported(self, xyz, features_pm):
            idx = ball_idx if ball_idx is not None else pointnet2_utils.ball_query(self.radius, self.nsample, xyz, new_xyz)
            if features_pm is None and features is not None:
                features_pm, feat_offset = features.transpose(1, 2).contiguous(), 0
            new_features, self.last_features_pm = fused_sa.sa_mlp_pool(
                self, xyz, new_xyz, idx, features_pm, feat_offset, inv=ball_inv)
            self.last_path = "fused"
            return new_xyz, new_features, inds
        if attention_blocks.get_backend() == "hip" and xyz.is_cuda:
            attention_blocks.fell_back("PointnetSAModuleVotes", self)
        grouped = self.grouper(xyz, new_xyz, features)
        if self.ret_unique_cnt:
            grouped_features, grouped_xyz, unique_cnt = grouped
        else:
            grouped_features, grouped_xyz = grouped
        new_features = self._pool(self.mlp_module(grouped_features), grouped_xyz)
        if self.ret_unique_cnt:
            return new_xyz, new_features, inds, unique_cnt
        return new_xyz, new_features, inds


class PointnetFPModule(nn.Module):
    """Inverse-distance 3-NN interpolation of ``known_feats`` onto ``unknown`` + skip concat + MLP.

    forward(unknown (B,n,3), known (B,m,3), unknow_feats (B,C1,n), known_feats (B,C2,m))
        -> (B, mlp[-1], n)                                  (pointnet2_modules.py:371-416)
    """

    def __init__(self, *, mlp, bn=True):
        super().__init__()
        self.mlp = SharedMLP(list(mlp), bn=bn)

    def forward(self, unknown, known, unknow_feats, known_feats, nn=None):
        """``nn`` (optional): (idx (B,n,3) i32, weight (B,n,3)) of the three nearest neighbours as
        ``Pointnet2Backbone.plan`` precomputed them."""
        from . import attention_blocks
        if known is not None and nn is not None:
            interpolated = pointnet2_utils.three_interpolate(known_feats.contiguous(), nn[0], nn[1])
        elif known is not None and attention_blocks.get_backend() == "hip" and unknown.is_cuda:
            # the same weights (same operations, same order) from ONE kernel instead of sqrt / add / reciprocal / sum / div
            from . import pointnet2_ext, rowwise
            dist2, idx = pointnet2_ext.three_nn(unknown.contiguous(), known.contiguous())
            weight = rowwise.three_nn_weights(dist2)
            interpolated = pointnet2_utils.three_interpolate(known_feats.contiguous(), idx, weight)
        elif known is not None:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            dist_recip = 1.0 / (dist + 1e-8)
            weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
            interpolated = pointnet2_utils.three_interpolate(known_feats.contiguous(), idx, weight)
        else:
            interpolated = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        if unknow_feats is not None:
            interpolated = torch.cat([interpolated, unknow_feats], dim=1)
        if attention_blocks.get_backend() == "hip" and interpolated.is_cuda and self._chain_ok():
            # SharedMLP = [1x1 conv -> BatchNorm2d -> ReLU] x n on position-major rows: grouped MFMA GEMMs
            # with the BatchNorm folded into the next operand load (fused_mlp); returned as the (B,C,n)
            # view of the position-major result, which is the layout the encoder wants next
            from .fused_mlp import mlp_chains
            b, c, n = interpolated.shape
            x_pm = interpolated.transpose(1, 2).reshape(b * n, c)
            out = mlp_chains(x_pm, [([(l.conv, l.bn.bn) for l in self.mlp], None, 0.0)], self.training)[0]
            self.last_path = "fused"
            return out.view(b, n, -1).transpose(1, 2)
        if attention_blocks.get_backend() == "hip" and interpolated.is_cuda:
            attention_blocks.fell_back("PointnetFPModule", self)
        return self.mlp(interpolated.unsqueeze(-1)).squeeze(-1)

    def _chain_ok(self):
        return all(hasattr(l, "bn") and l.conv.bias is None and l.conv.out_channels % 4 == 0
                   and hasattr(l, "activation") for l in self.mlp) and self.mlp[0].conv.in_channels % 4 == 0
