import sys; sys.path.insert(0,'.')
import torch
from butd_detr_amd import fused_attention as fa
def tg(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g):
            for _ in range(reps): fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
for M in (640, 2048, 8192):
    N=K=288
    x=torch.randn(M,K,device='cuda'); dy=torch.randn(M,N,device='cuda'); dw=torch.zeros(N,K,device='cuda'); db=torch.zeros(N,device='cuda')
    for split in (1,2,4,8,16,32,64):
        if split*32 > M: continue
        p = fa._problem(dy, x, dw, N, K, M, (1, N), (1, K), K, bias_grad=db, ones_col=True, accumulate=True, split_k=split)
        print(f"wgrad M={M} split={split}: {tg(lambda: fa._gemm([p], x)):.1f} us")
