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
out=[]
for (M,N,K) in [(2048,288,288),(8192,288,288),(640,288,288)]:
    x=torch.randn(M,K,device='cuda'); w=torch.randn(N,K,device='cuda'); b=torch.randn(N,device='cuda'); y=torch.empty(M,N,device='cuda')
    out.append(f"fwd {M}x{N}x{K}: {tg(lambda: fa._gemm([fa._fwd(x,w,y,M,N,K,bias=b)],x)):.1f}")
for (M,N,K) in [(1048576,128,64),(1048576,64,64),(262144,256,128),(262144,128,132)]:
    x=torch.randn(M,K,device='cuda'); w=torch.randn(N,K,device='cuda'); y=torch.empty(M,N,device='cuda')
    sc=torch.rand(K,device='cuda'); sh=torch.randn(K,device='cuda')
    t=tg(lambda: fa._gemm([fa._fwd(x,w,y,M,N,K,a_affine=(sc,sh))],x), 5)
    out.append(f"fwd+aff {M}x{N}x{K}: {t:.1f} us {2*M*N*K/t/1e6:.1f} TF")
    dy=torch.randn(M,N,device='cuda'); dw=torch.zeros(N,K,device='cuda'); dx=torch.empty(M,K,device='cuda')
    t=tg(lambda: fa._gemm([fa._wgrad(dy,x,dw,None,M,N,K,b_affine=(sc,sh)), fa._dgrad(dy,w,dx,M,N,K)],x), 5)
    out.append(f"wgrad+dgrad {M}x{N}x{K}: {t:.1f} us {4*M*N*K/t/1e6:.1f} TF")
print("\n".join(out))
