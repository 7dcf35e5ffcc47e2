import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from butd_detr_amd import pointnet2_ext as ext, _hiplib
from butd_detr_amd.synthetic_scenes import scene_batch
from oracle import pointnet2_oracle as orc
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pcs = np.ascontiguousarray(scene_batch(B, 1184, 50000)[..., :3])
d = torch.from_numpy(pcs).cuda()
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
got = ext.furthest_point_sampling(d, 2048)
ref = orc.furthest_point_sampling(pcs[:2], 2048, multithread=True)
print("pruned parity (2 scenes):", bool((got[:2].cpu().numpy() == ref).all()))
print("pruned FPS 50k->2048 B=%d: %.3f ms" % (B, timeit(lambda: ext.furthest_point_sampling(d, 2048))))
lib = _hiplib.load()
tmp = torch.empty(B, 50000, device='cuda'); out = torch.empty(B, 2048, dtype=torch.int32, device='cuda')
def stream_fn():
    lib.butd_furthest_point_sampling(B, 50000, 2048, d.data_ptr(), tmp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
print("streaming FPS: %.3f ms" % timeit(stream_fn, 2), " same:", bool((out == got).all()))
for n, m in [(2048, 1024), (1024, 512), (512, 256)]:
    x = d[:, :n].contiguous()
    print("FPS %d->%d: %.3f ms" % (n, m, timeit(lambda: ext.furthest_point_sampling(x, m))))
