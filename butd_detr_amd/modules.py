"""Query-generation and prediction-head modules between encoder and decoder (models/modules.py).

``PointsObjClsModule`` (:19-49), ``GeneralSamplingModule`` (:70-86), ``ThreeLayerMLP`` (:89-108) and
``ClsAgnosticPredictHead`` (:111-180); ``PositionEmbeddingLearned`` is shared with the
encoder/decoder file.  The Conv1d+BN1d+ReLU(+Dropout) stacks run as stock torch ops on the "torch"
backend and through the grouped-GEMM pipeline of include/butd_mlp.h (``fused_mlp.mlp_chains``) on the
"hip" backend: same parameters, same outputs, no NCHW round trips.
"""
import numpy as np
import torch.nn.functional as F
from torch import nn

from . import attention_blocks
from .encoder_decoder_layers import PositionEmbeddingLearned  # noqa: F401  (re-export, modules.py:52)
from .pointnet2_utils import gather_operation


def _fused(x):
    return attention_blocks.get_backend() == "hip" and x.is_cuda


def _position_major(features):
    """(B, C, L) -> (B*L, C) contiguous."""
    b, c, l = features.shape
    return features.transpose(1, 2).reshape(b * l, c)


class PointsObjClsModule(nn.Module):
    """Per-seed objectness logit: (B, C, K) -> (B, 1, K)."""

    def __init__(self, seed_feature_dim):
        super().__init__()
        self.in_dim = seed_feature_dim
        self.conv1 = nn.Conv1d(self.in_dim, self.in_dim, 1)
        self.bn1 = nn.BatchNorm1d(self.in_dim)
        self.conv2 = nn.Conv1d(self.in_dim, self.in_dim, 1)
        self.bn2 = nn.BatchNorm1d(self.in_dim)
        self.conv3 = nn.Conv1d(self.in_dim, 1, 1)

    def forward(self, seed_features, features_pm=None):
        """``features_pm``: optional (B, K, C) copy of the same features (skips one transpose)."""
        if _fused(seed_features):
            from .fused_mlp import mlp_chains
            b, _, k = seed_features.shape
            x = features_pm.reshape(b * k, -1) if features_pm is not None else _position_major(seed_features)
            out = mlp_chains(x, [([(self.conv1, self.bn1), (self.conv2, self.bn2)], self.conv3, 0.0)],
                             self.training)[0]
            return out.view(b, k, 1).transpose(1, 2)
        net = F.relu(self.bn1(self.conv1(seed_features)))
        net = F.relu(self.bn2(self.conv2(net)))
        return self.conv3(net)


class GeneralSamplingModule(nn.Module):
    """Gather xyz (B,K,3) and features (B,C,K) at ``sample_inds`` (B,Q) i32."""

    def forward(self, xyz, features, sample_inds):
        new_xyz = gather_operation(xyz.transpose(1, 2).contiguous(),
                                   sample_inds).transpose(1, 2).contiguous()
        new_features = gather_operation(features, sample_inds).contiguous()
        return new_xyz, new_features, sample_inds


class ThreeLayerMLP(nn.Module):
    """Conv1d-BN-ReLU-Dropout(0.3) x2 + Conv1d on (B, dim, N)."""

    def __init__(self, dim, out_dim):
        super().__init__()
        self.net = nn.Sequential(
            nn.Conv1d(dim, dim, 1, bias=False), nn.BatchNorm1d(dim), nn.ReLU(), nn.Dropout(0.3),
            nn.Conv1d(dim, dim, 1, bias=False), nn.BatchNorm1d(dim), nn.ReLU(), nn.Dropout(0.3),
            nn.Conv1d(dim, out_dim, 1))

    def chain(self):
        net = self.net
        return [(net[0], net[1]), (net[4], net[5])], net[8], net[3].p

    def forward(self, x):
        if _fused(x):
            from .fused_mlp import mlp_chains
            b, _, l = x.shape
            out = mlp_chains(_position_major(x), [self.chain()], self.training)[0]
            return out.view(b, l, -1).transpose(1, 2)
        return self.net(x)


class ClsAgnosticPredictHead(nn.Module):
    """Box centre / size / soft-token class heads writing ``{prefix}*`` into ``end_points``."""

    def __init__(self, num_class, num_heading_bin, num_proposal, seed_feat_dim=256,
                 objectness=True, heading=False, compute_sem_scores=True):
        super().__init__()
        self.num_class = num_class
        self.num_heading_bin = num_heading_bin
        self.num_proposal = num_proposal
        self.seed_feat_dim = seed_feat_dim
        self.objectness = objectness
        self.heading = heading
        self.compute_sem_scores = compute_sem_scores
        if objectness:
            self.objectness_scores_head = ThreeLayerMLP(seed_feat_dim, 1)
        self.center_residual_head = ThreeLayerMLP(seed_feat_dim, 3)
        if heading:
            self.heading_class_head = nn.Conv1d(seed_feat_dim, num_heading_bin, 1)
            self.heading_residual_head = nn.Conv1d(seed_feat_dim, num_heading_bin, 1)
        self.size_pred_head = ThreeLayerMLP(seed_feat_dim, 3)
        if compute_sem_scores:
            self.sem_cls_scores_head = ThreeLayerMLP(seed_feat_dim, self.num_class)

    def forward(self, features, base_xyz, end_points, prefix="", features_pm=None):
        """features (B, C, Q), base_xyz (B, Q, 3) -> (center (B,Q,3), pred_size (B,Q,3)).
        ``features_pm``: optional (B, Q, C) copy of the same features (the decoder's own layout)."""
        batch_size, num_proposal = features.shape[0], features.shape[-1]
        if _fused(features) and not self.heading:
            return self._forward_fused(features, base_xyz, end_points, prefix, features_pm)
        if self.objectness:
            scores = self.objectness_scores_head(features).transpose(2, 1)
            end_points[f"{prefix}objectness_scores"] = scores.squeeze(-1)
        center = base_xyz + self.center_residual_head(features).transpose(2, 1)
        if self.heading:
            heading_scores = self.heading_class_head(features).transpose(2, 1)
            normalized = self.heading_residual_head(features).transpose(2, 1)
            end_points[f"{prefix}heading_scores"] = heading_scores
            end_points[f"{prefix}heading_residuals_normalized"] = normalized
            end_points[f"{prefix}heading_residuals"] = normalized * (np.pi / self.num_heading_bin)
        pred_size = self.size_pred_head(features).transpose(2, 1).view(
            [batch_size, num_proposal, 3])
        end_points[f"{prefix}base_xyz"] = base_xyz
        end_points[f"{prefix}center"] = center
        end_points[f"{prefix}pred_size"] = pred_size
        if self.compute_sem_scores:
            end_points[f"{prefix}sem_cls_scores"] = self.sem_cls_scores_head(
                features).transpose(2, 1)
        return center, pred_size

    def _forward_fused(self, features, base_xyz, end_points, prefix, features_pm):
        """All ThreeLayerMLPs of the head as ONE group of chains on the position-major features."""
        from .fused_mlp import mlp_chains
        b, _, q = features.shape
        x = features_pm.reshape(b * q, -1) if features_pm is not None else _position_major(features)
        heads = ([self.objectness_scores_head] if self.objectness else []) + \
            [self.center_residual_head, self.size_pred_head] + \
            ([self.sem_cls_scores_head] if self.compute_sem_scores else [])
        outs = [o.view(b, q, -1) for o in mlp_chains(x, [h.chain() for h in heads], self.training)]
        if self.objectness:
            end_points[f"{prefix}objectness_scores"] = outs.pop(0).squeeze(-1)
        center = base_xyz + outs[0]
        pred_size = outs[1]
        end_points[f"{prefix}base_xyz"] = base_xyz
        end_points[f"{prefix}center"] = center
        end_points[f"{prefix}pred_size"] = pred_size
        if self.compute_sem_scores:
            end_points[f"{prefix}sem_cls_scores"] = outs[2]
        return center, pred_size
