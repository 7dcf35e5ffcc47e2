"""round 4: does the evaluation order of BatchNorm -- one fma `z * scale + shift` (the fused path, and
ATen's own CPU kernel: aten/src/ATen/native/cpu/batch_norm_kernel.cpp computes alpha = invstd * weight,
beta = bias - mean * alpha, out = in * alpha + beta) versus `(z - mean) * rstd * gamma + beta` -- explain the ReLU-gate
flips against the CPU-generated reference vectors?  CPU only.  Channels with |mean| up to 60 x std."""
import torch
torch.manual_seed(0)
N, C = 200000, 64
z = (torch.randn(N, C) * 0.5 + torch.linspace(-30, 30, C))
bn = torch.nn.BatchNorm1d(C).train()
with torch.no_grad():
    bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.2, 0.2)
y = bn(z)
mean, var = z.double().mean(0), z.double().var(0, unbiased=False)
m32, v32 = z.mean(0), z.var(0, unbiased=False)
print(f"stock torch CPU BatchNorm1d (train mode) on {N} x {C}, channel means -30 .. 30, std 0.5; compared with:")
for name, (mu, va) in {"fp64 statistics": (mean, var), "fp32 statistics": (m32.double(), v32.double())}.items():
    rstd = 1 / torch.sqrt(va + bn.eps)
    scale = (bn.weight.double() * rstd).float(); shift = (bn.bias.double() - mu * scale.double()).float()
    a = torch.addcmul(shift, z, scale)
    b = ((z - mu.float()) * rstd.float()) * bn.weight + bn.bias
    for nm, t in (("z * scale + shift", a), ("(z - mean) * rstd * gamma + beta", b)):
        d = (t - y).abs()
        print(f"  {name:16s} {nm:34s} max|diff| {d.max():.3e}  mean {d.mean():.3e}  bit-equal {(t == y).float().mean():.4f}"
              f"  gate flips {((t > 0) != (y > 0)).sum().item()} of {t.numel()}")
