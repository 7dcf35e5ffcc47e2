import sys; sys.path.insert(0,'.')
import torch
from butd_detr_amd import fused_attention as fa
def tg(fn, reps=50):
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
for (M,N,K) in [(2048,288,288),(2048,288,16),(2048,64,288),(8192,288,288),(8192,288,16),(640,288,288),(128,64,64)]:
    x=torch.randn(M,K,device='cuda'); w=torch.randn(N,K,device='cuda'); b=torch.randn(N,device='cuda'); y=torch.empty(M,N,device='cuda')
    print(f"M={M} N={N} K={K}: mine {tg(lambda: fa._gemm([fa._fwd(x,w,y,M,N,K,bias=b)],x)):.1f} us   torch addmm {tg(lambda: torch.addmm(b,x,w.t(),out=y)):.1f} us")
z=torch.zeros(16,device='cuda')
print("tiny torch op:", tg(lambda: z.add_(1)))
