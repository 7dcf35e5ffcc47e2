"""plain fp32 products: torch.mm (hipBLASLt / rocBLAS) vs butd_gemm_grouped on the step's common shapes (graph replay)"""
import sys; sys.path.insert(0, '.')
import torch
from butd_detr_amd import fused_attention as fa
torch.backends.cuda.matmul.allow_tf32 = False
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
for M, N, K in ((2048, 288, 288), (8192, 288, 288), (8192, 576, 288), (14336, 288, 288), (65536, 128, 128), (262144, 256, 128), (1048576, 64, 64), (1048576, 128, 64)):
    x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda'); y = torch.empty(M, N, device='cuda')
    t_lib = tg(lambda: torch.mm(x, w.t(), out=y))
    t_own = tg(lambda: fa._gemm([fa._fwd(x, w, y, M, N, K)], x))
    fl = 2.0 * M * N * K
    print(f"{M:8d} x {N:3d} x {K:3d}: library {t_lib:7.1f} us ({fl / t_lib / 1e6:5.1f} TF)   butd_gemm_grouped {t_own:7.1f} us ({fl / t_own / 1e6:5.1f} TF)")
# weight-gradient shape: (N x M) @ (M x K)
for M, N, K in ((2048, 288, 288), (8192, 288, 288), (262144, 256, 128)):
    dy = torch.randn(M, N, device='cuda'); x = torch.randn(M, K, device='cuda'); dw = torch.zeros(N, K, device='cuda')
    t_lib = tg(lambda: torch.mm(dy.t(), x, out=dw))
    t_own = tg(lambda: fa._gemm([fa._wgrad(dy, x, dw, None, M, N, K)], x))
    fl = 2.0 * M * N * K
    print(f"wgrad {N:3d} x {K:3d} x {M:7d}: library {t_lib:7.1f} us ({fl / t_lib / 1e6:5.1f} TF)   butd_gemm_grouped {t_own:7.1f} us ({fl / t_own / 1e6:5.1f} TF)")
