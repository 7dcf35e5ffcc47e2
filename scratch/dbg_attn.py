import sys; sys.path.insert(0,'.')
import torch, math
from butd_detr_amd import fused_attention as fa, _hiplib
lib = _hiplib.load()
torch.manual_seed(0)
def core(B,H,Lq,Lk,D,mask=None):
    E=H*D
    q=torch.randn(B,Lq,E,device='cuda'); k=torch.randn(B,Lk,E,device='cuda'); v=torch.randn(B,Lk,E,device='cuda')
    o=torch.empty(B,Lq,E,device='cuda'); lse=torch.empty(B,H,Lq,device='cuda')
    err=lib.butd_attention_fwd(B,H,Lq,Lk,D,q.data_ptr(),k.data_ptr(),v.data_ptr(),None if mask is None else mask.data_ptr(),o.data_ptr(),lse.data_ptr(),0.0,0,None,torch.cuda.current_stream().cuda_stream)
    assert err==0
    qh=q.view(B,Lq,H,D).transpose(1,2); kh=k.view(B,Lk,H,D).transpose(1,2); vh=v.view(B,Lk,H,D).transpose(1,2)
    s=qh@kh.transpose(-1,-2)
    if mask is not None: s=s.masked_fill(mask[:,None,None,:],float('-inf'))
    ref=(torch.softmax(s,-1)@vh).transpose(1,2).reshape(B,Lq,E)
    print(f"core B{B} Lq{Lq} Lk{Lk} mask={mask is not None}: max err {(o-ref).abs().max().item():.2e}  lse err {(lse-torch.logsumexp(s,-1)).abs().max().item():.2e}")
core(1,8,64,64,36); core(1,8,64,80,36); core(1,8,64,16,36); core(2,8,128,80,36)
m=torch.zeros(2,80,dtype=torch.bool,device='cuda'); m[0,79:]=True; m[1,70:]=True
core(2,8,128,80,36,m)
m=torch.zeros(3,80,dtype=torch.bool,device='cuda'); m[0,79:]=True; m[1,78:]=True; m[2,77:]=True
core(3,8,1024,80,36,m)
# gemm check
def gemm(M,N,K):
    x=torch.randn(M,K,device='cuda'); w=torch.randn(N,K,device='cuda'); b=torch.randn(N,device='cuda'); y=torch.empty(M,N,device='cuda')
    fa._gemm([fa._fwd(x,w,y,M,N,K,bias=b,scale=0.5)],x)
    ref=(x@w.t()+b)*0.5
    print(f"gemm {M}x{N}x{K}: {(y-ref).abs().max().item():.2e}")
gemm(64,64,16); gemm(240,288,288); gemm(3072,288,288); gemm(100,36,52)
x=torch.randn(240,288,device='cuda'); x2=torch.randn(3072,288,device='cuda'); w=torch.randn(864,288,device='cuda'); b=torch.randn(864,device='cuda')
q=torch.empty(3072,288,device='cuda'); k=torch.empty(240,288,device='cuda'); v=torch.empty(240,288,device='cuda')
fa._gemm([fa._fwd(x2,w[:288],q,3072,288,288,bias=b[:288],scale=0.3), fa._fwd(x,w[288:576],k,240,288,288,bias=b[288:576]), fa._fwd(x,w[576:],v,240,288,288,bias=b[576:])],x)
print("grouped:", ((x2@w[:288].t()+b[:288])*0.3-q).abs().max().item(), ((x@w[288:576].t()+b[288:576])-k).abs().max().item(), ((x@w[576:].t()+b[576:])-v).abs().max().item())
