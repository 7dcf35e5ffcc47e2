import sys; sys.path.insert(0,'.')
import torch
from butd_detr_amd import fused_attention as fa, _hiplib
lib=_hiplib.load()
import os
BF=os.environ.get('BF16')=='1'
FWD=lib.butd_attention_fwd_bf16 if BF else lib.butd_attention_fwd
BWD=lib.butd_attention_bwd_bf16 if BF else lib.butd_attention_bwd
B,H,D=8,8,36; E=H*D; Lq=Lk=1024
q=torch.randn(B,Lq,E,device='cuda'); k=torch.randn(B,Lk,E,device='cuda'); v=torch.randn(B,Lk,E,device='cuda')
out=torch.empty_like(q); lse=torch.empty(B,H,Lq,device='cuda'); do=torch.randn_like(q)
dq=torch.empty_like(q); dk=torch.empty_like(k); dv=torch.empty_like(v); delta=torch.empty(B,H,Lq,device='cuda')
ctr=fa.rng_counter(q.device).data_ptr(); s=torch.cuda.current_stream().cuda_stream
for _ in range(5):
    FWD(B,H,Lq,Lk,D,q.data_ptr(),k.data_ptr(),v.data_ptr(),None,out.data_ptr(),lse.data_ptr(),0.1,7,ctr,s)
    BWD(B,H,Lq,Lk,D,q.data_ptr(),k.data_ptr(),v.data_ptr(),None,out.data_ptr(),do.data_ptr(),lse.data_ptr(),delta.data_ptr(),dq.data_ptr(),dk.data_ptr(),dv.data_ptr(),0,0,1.0,0.1,7,ctr,s)
torch.cuda.synchronize()
