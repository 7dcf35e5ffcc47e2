"""SA1 forward in eval mode at the bench size (8 x 50 000 points, 2048 centres, 64 neighbours): the one-kernel
level vs the multi-launch pipeline -- time per forward (graph replay) ; run under rocprofv3 --pmc for HBM bytes."""
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
levels = [dict(npoint=2048, radius=0.2, nsample=64, mlp=[3, 64, 64, 128])]
m = PointnetSAModuleVotes(use_xyz=True, normalize_xyz=True, **levels[0]).to(dev).eval()
inds = pointnet2_utils.furthest_point_sample(xyz, 2048)
new_xyz = torch.gather(xyz, 1, inds.long()[..., None].expand(-1, -1, 3)).contiguous()
idx = pointnet2_utils.ball_query(0.2, 64, xyz, new_xyz)
def tg(fn, reps=5):
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
with torch.no_grad():
    one = lambda: fused_sa.sa_fused_eval(m, xyz, new_xyz, idx, pc, 3)
    fused_sa._FUSED_EVAL[0] = False
    many = lambda: fused_sa.sa_mlp_pool(m, xyz, new_xyz, idx, pc, 3)
    a, b = one()[0], many()[0]
    print("max |one - many| / scale:", float((a - b).abs().max() / b.abs().max()))
    if len(sys.argv) > 1 and sys.argv[1] == "pmc":
        one(); many(); torch.cuda.synchronize()
    else:
        print("one kernel   : %.1f us" % tg(one))
        print("multi-launch : %.1f us" % tg(many))
