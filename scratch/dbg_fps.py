import numpy as np, torch, sys
sys.path.insert(0, '.')
from butd_detr_amd import pointnet2_ext as ext
from oracle import pointnet2_oracle as orc
for n, m in [(7, 7), (64, 10), (100, 64), (2048, 64)]:
    rng = np.random.default_rng(n * 31 + m)
    pts = rng.uniform(-2, 2, size=(1, n, 3)).astype(np.float32)
    got = ext.furthest_point_sampling(torch.from_numpy(pts).cuda(), m).cpu().numpy()
    ref = orc.furthest_point_sampling(pts, m)
    print(n, m, "match" if (got == ref).all() else "MISMATCH", got[0][:12], ref[0][:12])
