import torch
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
x=torch.randn(1048576,128,device='cuda'); y=torch.empty_like(x); z=torch.randn_like(x)
nb=x.numel()*4
print("mul_ in place   : %.1f us  %.2f TB/s" % ((u:=t(lambda: x.mul_(1.0001))), 2*nb/u/1e6))
print("copy            : %.1f us  %.2f TB/s" % ((u:=t(lambda: y.copy_(x))), 2*nb/u/1e6))
print("add out         : %.1f us  %.2f TB/s" % ((u:=t(lambda: torch.add(x,z,out=y))), 3*nb/u/1e6))
print("sum (read only) : %.1f us  %.2f TB/s" % ((u:=t(lambda: x.sum())), nb/u/1e6))
print("fill            : %.1f us  %.2f TB/s" % ((u:=t(lambda: y.fill_(1.0))), nb/u/1e6))
