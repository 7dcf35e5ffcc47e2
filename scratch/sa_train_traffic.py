"""SA1 / SA2 in TRAINING mode at the bench size (8 x 50 000 points): one forward + backward of the multi-launch pipeline
(fused_sa.sa_mlp_pool), per-kernel list for rocprofv3 --pmc (HBM bytes) or, without arguments, graph-replay times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from butd_detr_amd import attention_blocks, fused_sa, pointnet2_utils
from butd_detr_amd.pointnet2_modules import PointnetSAModuleVotes
from butd_detr_amd.train_step import synthetic_batch
attention_blocks.set_backend("hip")
dev = torch.device("cuda", 0)
inputs, _ = synthetic_batch(8, dev, seed=1184, n_points=50000, tokens=80)
pc = inputs["point_clouds"]; xyz = pc[..., :3].contiguous()
level = int(os.environ.get("LEVEL", "1"))
if level == 1:
    m = PointnetSAModuleVotes(use_xyz=True, normalize_xyz=True, npoint=2048, radius=0.2, nsample=64, mlp=[3, 64, 64, 128]).to(dev).train()
    src_xyz, feats_pm, off = xyz, pc, 3
else:
    inds1 = pointnet2_utils.furthest_point_sample(xyz, 2048)
    src_xyz = torch.gather(xyz, 1, inds1.long()[..., None].expand(-1, -1, 3)).contiguous()
    feats_pm, off = torch.randn(8, 2048, 128, device=dev), 0
    m = PointnetSAModuleVotes(use_xyz=True, normalize_xyz=True, npoint=1024, radius=0.4, nsample=32, mlp=[128, 128, 128, 256]).to(dev).train()
np_ = m.npoint
inds = pointnet2_utils.furthest_point_sample(src_xyz, np_)
new_xyz = torch.gather(src_xyz, 1, inds.long()[..., None].expand(-1, -1, 3)).contiguous()
idx = pointnet2_utils.ball_query(m.radius, m.nsample, src_xyz, new_xyz)
feats_pm = feats_pm.clone().requires_grad_(level != 1)
probe = torch.randn(8, np_, m.mlp_module[-1].conv.out_channels, device=dev)
inv = fused_sa.inverse_index(idx, src_xyz.shape[1]) if level != 1 else None      # (coordinates only: from the prefetched plan in the step)
def step():
    for p in m.parameters(): p.grad = None
    out_cm, out_pm = fused_sa.sa_mlp_pool(m, src_xyz, new_xyz, idx, feats_pm, off, inv=inv)
    (out_pm * probe).sum().backward()
def tg(fn, reps=3):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g):
            for _ in range(reps): fn()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
if len(sys.argv) > 1 and sys.argv[1] == "pmc":
    step(); step(); torch.cuda.synchronize()
else:
    print("SA%d training forward + backward: %.1f us" % (level, tg(step)))
