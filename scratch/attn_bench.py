import sys; sys.path.insert(0,'.')
import torch, math
from butd_detr_amd import fused_attention as fa, _hiplib
lib=_hiplib.load()
import os
BF=os.environ.get('BF16')=='1'   # BF16=1: the bf16 entry points (bf16 LDS images, round 5)
FWD=lib.butd_attention_fwd_bf16 if BF else lib.butd_attention_fwd
BWD=lib.butd_attention_bwd_bf16 if BF else lib.butd_attention_bwd
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
B,H,D=8,8,36; E=H*D
for Lq,Lk in ((1024,1024),(256,1024),(256,256),(256,80),(1024,80),(80,1024),(1024,132),(256,132)):
    q=torch.randn(B,Lq,E,device='cuda'); k=torch.randn(B,Lk,E,device='cuda'); v=torch.randn(B,Lk,E,device='cuda')
    out=torch.empty_like(q); lse=torch.empty(B,H,Lq,device='cuda'); do=torch.randn_like(q)
    dq=torch.empty_like(q); dk=torch.empty_like(k); dv=torch.empty_like(v); delta=torch.empty(B,H,Lq,device='cuda')
    ctr=fa.rng_counter(q.device).data_ptr(); st=lambda: torch.cuda.current_stream().cuda_stream
    for p in (0.0, 0.1):
        f=lambda: FWD(B,H,Lq,Lk,D,q.data_ptr(),k.data_ptr(),v.data_ptr(),None,out.data_ptr(),lse.data_ptr(),p,7,ctr,st())
        b=lambda: BWD(B,H,Lq,Lk,D,q.data_ptr(),k.data_ptr(),v.data_ptr(),None,out.data_ptr(),do.data_ptr(),lse.data_ptr(),delta.data_ptr(),dq.data_ptr(),dk.data_ptr(),dv.data_ptr(),0,0,1.0,p,7,ctr,st())
        print(f"Lq={Lq} Lk={Lk} p={p}: fwd {tg(f):.1f} us  bwd {tg(b):.1f} us")

