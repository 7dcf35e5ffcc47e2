import sys; sys.path.insert(0,'.')
import torch, math
from butd_detr_amd import _hiplib
lib = _hiplib.load()
torch.manual_seed(0)
B,H,Lq,Lk,D=1,1,16,16,36
E=H*D
q=torch.randn(B,Lq,E,device='cuda')*0.1; k=torch.randn(B,Lk,E,device='cuda')*0.1; v=torch.arange(Lk,device='cuda',dtype=torch.float32)[None,:,None].expand(B,Lk,E).contiguous()
for mk in ([15],[0],[3,7],list(range(8,16))):
    mask=torch.zeros(B,Lk,dtype=torch.bool,device='cuda'); mask[0,mk]=True
    o=torch.empty(B,Lq,E,device='cuda'); lse=torch.empty(B,H,Lq,device='cuda')
    err=lib.butd_attention_fwd(B,H,Lq,Lk,D,q.data_ptr(),k.data_ptr(),v.data_ptr(),mask.data_ptr(),o.data_ptr(),lse.data_ptr(),0.0,0,None,torch.cuda.current_stream().cuda_stream)
    s=(q@k.transpose(-1,-2)).masked_fill(mask[:,None,:],float('-inf'))
    ref=torch.softmax(s,-1)@v
    print(mk, "out[0,:4,0]", o[0,:4,0].tolist(), "ref", ref[0,:4,0].tolist(), "lse", lse[0,0,:2].tolist(), torch.logsumexp(s,-1)[0,:2].tolist())
