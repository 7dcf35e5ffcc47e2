"""add + dropout + LayerNorm forward / backward kernels alone (graph replay of 20 launches), vs their HBM bytes"""
import sys; sys.path.insert(0, '.')
import torch
from butd_detr_amd import _hiplib
lib = _hiplib.load()
def tg(fn, reps=20):
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
for rows in (8192, 2048, 640):
    cols = 288
    dy, x, res = (torch.randn(rows, cols, device="cuda") for _ in range(3))
    g = torch.ones(cols, device="cuda"); b = torch.zeros(cols, device="cuda")
    mean = torch.zeros(rows, device="cuda"); rstd = torch.ones(rows, device="cuda")
    y, dx, dres = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    dg, db = torch.zeros(cols, device="cuda"), torch.zeros(cols, device="cuda")
    ctr = torch.zeros(1, dtype=torch.int64, device="cuda")
    st = lambda: torch.cuda.current_stream().cuda_stream
    for p in (0.0, 0.1):
        fw = lambda: lib.butd_add_dropout_layernorm_fwd(rows, cols, x.data_ptr(), res.data_ptr(), g.data_ptr(), b.data_ptr(), 1e-5, y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), p, 5, ctr.data_ptr(), st())
        bw = lambda: lib.butd_add_dropout_layernorm_bwd(rows, cols, dy.data_ptr(), x.data_ptr(), res.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), dres.data_ptr(), dg.data_ptr(), db.data_ptr(), p, 5, ctr.data_ptr(), st())
        mb = rows * cols * 4 / 1e6
        tf, tb = tg(fw), tg(bw)
        print(f"rows={rows} p={p}: fwd {tf:6.1f} us ({3 * mb / tf * 1e-3 * 1e3:5.0f} GB/s over 3 tensors)  bwd {tb:6.1f} us ({5 * mb / tb * 1e-3 * 1e3:5.0f} GB/s over 5 tensors)")
