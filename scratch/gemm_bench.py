import sys; sys.path.insert(0,'.')
import torch
from butd_detr_amd import fused_attention as fa
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
for M in (8192, 2048, 640):
    N=K=288
    x=torch.randn(M,K,device='cuda'); w=torch.randn(N,K,device='cuda'); b=torch.randn(N,device='cuda'); y=torch.empty(M,N,device='cuda')
    dy=torch.randn(M,N,device='cuda'); dx=torch.empty(M,K,device='cuda'); dw=torch.zeros(N,K,device='cuda'); db=torch.zeros(N,device='cuda')
    print(f"M={M}: fwd {t(lambda: fa._gemm([fa._fwd(x,w,y,M,N,K,bias=b)],x)):.1f} us (torch addmm {t(lambda: torch.addmm(b,x,w.t())):.1f}) | "
          f"fwd x3 grouped {t(lambda: fa._gemm([fa._fwd(x,w,y,M,N,K,bias=b)]*3,x)):.1f} | "
          f"dgrad {t(lambda: fa._gemm([fa._dgrad(dy,w,dx,M,N,K)],x)):.1f} (torch mm {t(lambda: dy@w):.1f}) | "
          f"wgrad {t(lambda: fa._gemm([fa._wgrad(dy,x,dw,db,M,N,K)],x)):.1f} (torch {t(lambda: dy.t()@x):.1f})")
