"""ball query: streaming kernel vs grid-pruned path at the four SA levels of configs[1] (HIP events)."""
import os
import numpy as np, torch
from butd_detr_amd import _hiplib, pointnet2_ext as ext
from butd_detr_amd.synthetic_scenes import scene_batch
lib = _hiplib.load()
pcs = torch.from_numpy(np.ascontiguousarray(scene_batch(8, 1184, 50000)[..., :3])).cuda()
s = torch.cuda.current_stream().cuda_stream
def timeit(f, it=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(it): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
xyz = pcs
for npoint, r, ns in ((2048, 0.2, 64), (1024, 0.4, 32), (512, 0.8, 16), (256, 1.2, 16)):
    inds = ext.furthest_point_sampling(xyz, npoint)
    new_xyz = torch.gather(xyz, 1, inds.long()[..., None].expand(-1, -1, 3)).contiguous()
    b, n, m = xyz.shape[0], xyz.shape[1], npoint
    idx = torch.empty((b, m, ns), dtype=torch.int32, device="cuda")
    idx2 = torch.empty_like(idx)
    ws = torch.empty(24 * b * n + 264192 * b, dtype=torch.uint8, device="cuda")
    t0 = timeit(lambda: lib.butd_ball_query(b, n, m, r, ns, new_xyz.data_ptr(), xyz.data_ptr(), idx.data_ptr(), s))
    t1 = timeit(lambda: lib.butd_ball_query_ws(b, n, m, r, ns, new_xyz.data_ptr(), xyz.data_ptr(), idx2.data_ptr(), ws.data_ptr(), ws.numel(), s))
    hits = (idx != idx[..., :1]).sum(-1).float().mean().item() + 1
    print(f"n={n} m={m} r={r} ns={ns}: streaming {t0:.1f} us  pruned {t1:.1f} us  equal={torch.equal(idx, idx2)} ~hits {hits:.1f} wsbytes={lib.butd_ball_query_workspace_bytes(b,n,m)}")
    xyz = new_xyz
    if os.environ.get('BQ_ONLY_L1'): raise SystemExit
for b1 in (1, 2):
    x = pcs[:b1].contiguous(); inds = ext.furthest_point_sampling(x, 2048)
    c = torch.gather(x, 1, inds.long()[..., None].expand(-1, -1, 3)).contiguous()
    idx = torch.empty((b1, 2048, 64), dtype=torch.int32, device="cuda")
    ws = torch.empty(24 * b1 * 50000 + 264192 * b1, dtype=torch.uint8, device="cuda")
    t0 = timeit(lambda: lib.butd_ball_query(b1, 50000, 2048, 0.2, 64, c.data_ptr(), x.data_ptr(), idx.data_ptr(), s))
    t1 = timeit(lambda: lib.butd_ball_query_ws(b1, 50000, 2048, 0.2, 64, c.data_ptr(), x.data_ptr(), idx.data_ptr(), ws.data_ptr(), ws.numel(), s))
    print(f"b={b1}: streaming {t0:.1f} us pruned {t1:.1f} us")
