import sys; sys.path.insert(0,'.')
import torch
from butd_detr_amd import fused_attention as fa
def tg(fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g):
            for _ in range(reps): fn()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
M,N,K=1048576,64,8
x=torch.randn(M,K,device='cuda'); dy=torch.randn(M,N,device='cuda'); dw=torch.zeros(N,K,device='cuda'); w=torch.randn(N,K,device='cuda'); y=torch.empty(M,N,device='cuda')
print(f"fwd (1M,64,8): {tg(lambda: fa._gemm([fa._fwd(x,w,y,M,N,K)],x)):.1f} us")
for split in (64,128,256,512,1024):
    p = fa._problem(dy, x, dw, N, K, M, (1, N), (1, K), K, accumulate=True, split_k=split)
    print(f"wgrad (64,8,1M) split={split}: {tg(lambda: fa._gemm([p], x)):.1f} us")
for (N,K,M,splits) in ((64,64,1048576,(128,256,512,1024)),(128,64,1048576,(128,256,512)),(256,128,262144,(64,128,256,512)),(128,128,262144,(64,128,256,512)),(128,132,262144,(64,128,256,512))):
    x=torch.randn(M,K,device='cuda'); dy=torch.randn(M,N,device='cuda'); dw=torch.zeros(N,K,device='cuda')
    for split in splits:
        p = fa._problem(dy, x, dw, N, K, M, (1, N), (1, K), K, accumulate=True, split_k=split)
        print(f"wgrad ({N},{K},{M}) split={split}: {tg(lambda: fa._gemm([p], x)):.1f} us")
