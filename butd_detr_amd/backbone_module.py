"""PointNet++ single-scale-grouping backbone of BUTD-DETR (models/backbone_module.py:35-153).

Four set-abstraction levels (2048/0.2/64, 1024/0.4/32, 512/0.8/16, 256/1.2/16; all with
``use_xyz`` and radius-normalised local xyz) and two feature-propagation levels back up to the
1024 seed points; emits the ``sa{1..4}_*`` and ``fp2_*`` entries of ``end_points``.
"""
import torch
from torch import nn

from . import pointnet2_utils
from .pointnet2_modules import PointnetFPModule, PointnetSAModuleVotes

# (npoint, radius, nsample, hidden width, output width) per level -- backbone_module.py:53-87
SA_LEVELS = ((2048, 0.2, 64, 64, 128), (1024, 0.4, 32, 128, 256),
             (512, 0.8, 16, 128, 256), (256, 1.2, 16, 128, 256))


class Pointnet2Backbone(nn.Module):
    def __init__(self, input_feature_dim=0, width=1, depth=2, output_dim=288):
        super().__init__()
        self.depth, self.width = depth, width
        c_in = input_feature_dim
        for level, (npoint, radius, nsample, hidden, c_out) in enumerate(SA_LEVELS, start=1):
            widths = [c_in] + [hidden * width] * depth + [c_out * width]
            setattr(self, f"sa{level}", PointnetSAModuleVotes(
                npoint=npoint, radius=radius, nsample=nsample, mlp=widths,
                use_xyz=True, normalize_xyz=True))
            c_in = c_out * width
        self.fp1 = PointnetFPModule(mlp=[512 * width, 256 * width, 256 * width])
        self.fp2 = PointnetFPModule(mlp=[512 * width, 256 * width, output_dim])

    @staticmethod
    def _break_up_pc(pc):
        xyz = pc[..., 0:3].contiguous()
        features = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
        return xyz, features

    @torch.no_grad()
    def sample(self, pointcloud):
        """The furthest-point-sampling chain of the four levels on its own: [inds1 (B,2048), inds2
        (B,1024), inds3 (B,512), inds4 (B,256)] int32.  It depends on the coordinates only -- not on any
        parameter -- so a training loop can run it for batch k+1 (a long serial kernel on 8 CUs) while
        batch k trains, and hand the result to ``forward(..., sample_inds=...)``."""
        xyz = pointcloud[..., 0:3].contiguous()
        out = []
        for level in (1, 2, 3, 4):
            sa = getattr(self, f"sa{level}")
            inds = pointnet2_utils.furthest_point_sample(xyz, sa.npoint)
            out.append(inds)
            xyz = pointnet2_utils.gather_operation(
                xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
        return out

    @torch.no_grad()
    def plan(self, pointcloud):
        """Everything of the backbone that depends on the COORDINATES alone -- no parameter, no feature: per level the
        furthest-point samples, the sampled centres and the ball-query neighbour lists (pointnet2_modules.py:225-241,
        pointnet2_utils.py:346-349), and the three-nearest-neighbour indices + inverse-distance weights of the two
        feature-propagation modules (pointnet2_modules.py:392-396).  A training loop runs it for batch k+1 on a side
        stream while batch k trains and hands it to ``forward(..., plan=...)``:
        {"levels": [(inds (B,np) i32, new_xyz (B,np,3), idx (B,np,ns) i32, inverse lists (start, list) or None) x 4],
         "fp": [(idx (B,n,3) i32, weight (B,n,3)) x 2]}"""
        from . import fused_sa, pointnet2_ext, rowwise
        xyz = pointcloud[..., 0:3].contiguous()
        xyzs, levels = [xyz], []
        for level in (1, 2, 3, 4):
            sa = getattr(self, f"sa{level}")
            inds = pointnet2_utils.furthest_point_sample(xyz, sa.npoint)
            new_xyz = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
            idx = pointnet2_utils.ball_query(sa.radius, sa.nsample, xyz, new_xyz)
            # levels 2-4 return a feature gradient: the inverted neighbour lists turn its scatter into a gather
            inv = fused_sa.inverse_index(idx, xyz.shape[1]) if (level > 1 and idx.is_cuda) else None
            levels.append((inds, new_xyz, idx, inv))
            xyz = new_xyz
            xyzs.append(xyz)
        fp = []
        for unknown, known in ((xyzs[3], xyzs[4]), (xyzs[2], xyzs[3])):     # fp1: sa3 <- sa4, fp2: sa2 <- sa3 (:139-147)
            dist2, nn_idx = pointnet2_ext.three_nn(unknown, known)
            fp.append((nn_idx, rowwise.three_nn_weights(dist2)))
        return {"levels": levels, "fp": fp}

    def forward(self, pointcloud, end_points=None, sample_inds=None, plan=None):
        """pointcloud (B, N, 3 + input_feature_dim) -> end_points dict.  ``sample_inds``: optional
        precomputed result of ``sample(pointcloud)`` (the modules' own ``inds`` argument,
        pointnet2_modules.py:210-241); ``plan``: optional result of ``plan(pointcloud)`` (samples, centres, neighbour
        lists, interpolation weights: the levels then start at their grouping)."""
        end_points = end_points if end_points else {}
        xyz, features = self._break_up_pc(pointcloud)
        # point-major hand-over between levels (used by the fused gfx950 SA pipeline, ignored otherwise)
        feats_pm, off = (pointcloud.contiguous(), 3) if features is not None else (None, 0)
        for level in (1, 2, 3, 4):
            sa = getattr(self, f"sa{level}")
            if plan is not None:
                p_inds, p_xyz, p_idx, p_inv = (tuple(plan["levels"][level - 1]) + (None,))[:4]
                xyz, features, inds = sa(xyz, features, features_pm=feats_pm, feat_offset=off, inds=p_inds,
                                         new_xyz=p_xyz, ball_idx=p_idx, ball_inv=p_inv)
            else:
                xyz, features, inds = sa(xyz, features, features_pm=feats_pm, feat_offset=off,
                                         inds=None if sample_inds is None else sample_inds[level - 1])
            feats_pm, off = sa.last_features_pm, 0
            # (a module attribute that holds a tensor with a grad_fn keeps this iteration's autograd graph -- and the
            #  parameters' AccumulateGrad nodes with the stream they were created on -- alive into the next one)
            sa.last_features_pm = None
            if level <= 2:  # the reference only records inds of the first two levels (:127,:132)
                end_points[f"sa{level}_inds"] = inds
            end_points[f"sa{level}_xyz"] = xyz
            end_points[f"sa{level}_features"] = features
        nn1, nn2 = plan["fp"] if plan is not None else (None, None)
        features = self.fp1(end_points["sa3_xyz"], end_points["sa4_xyz"],
                            end_points["sa3_features"], end_points["sa4_features"], nn=nn1)
        features = self.fp2(end_points["sa2_xyz"], end_points["sa3_xyz"],
                            end_points["sa2_features"], features, nn=nn2)
        end_points["fp2_features"] = features
        end_points["fp2_xyz"] = end_points["sa2_xyz"]
        num_seed = end_points["fp2_xyz"].shape[1]
        # seeds are a prefix of the sa1 samples: indices into the full input cloud (:150-151)
        end_points["fp2_inds"] = end_points["sa1_inds"][:, 0:num_seed]
        return end_points
