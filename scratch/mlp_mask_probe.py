"""Two ThreeLayerMLP chains (G = 2) with Dropout 0.3 through fused_mlp.mlp_chains: forward and backward against a torch
restatement handed the masks butd_mlp_mask_stats returns, with the gate/statistics epilogue on and off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from butd_detr_amd import _hiplib, fused_attention as fa, fused_mlp
from butd_detr_amd.modules import ThreeLayerMLP
torch.manual_seed(0)
dev = torch.device("cuda", 0)
lib = _hiplib.load()
OUTS = [int(v) for v in os.environ.get('OUTS', '7,7').split(',')]
G, H, P = len(OUTS), 288, int(os.environ.get('ROWS', '800'))
mlps = [ThreeLayerMLP(H, n).cuda().train() for n in OUTS]
state = [{n: b.clone() for n, b in m.named_buffers()} for m in mlps]
x = torch.randn(P, H, device=dev, requires_grad=True)
probes = [torch.randn(P, n, device=dev) for n in OUTS]
fa.new_step(dev)
def run(fuse):
    fused_mlp.set_fuse_stats(fuse)
    with torch.no_grad():
        for m, st in zip(mlps, state):
            for n, b in m.named_buffers(): b.copy_(st[n])
    fa._site[0] = 300
    x.grad = None
    for m in mlps: m.zero_grad()
    ys = fused_mlp.mlp_chains(x, [m.chain() for m in mlps], True)
    sum((y * p).sum() for y, p in zip(ys, probes)).backward()
    return [y.detach().clone() for y in ys], [x.grad.clone()] + [p.grad.clone() for m in mlps for p in m.parameters()]
masks = []
for l in range(2):
    m = torch.ones(P, G * H, device=dev)
    zeros, ones = torch.zeros(P, G * H, device=dev), torch.ones(G * H, device=dev)
    S = torch.zeros(2, G * H, dtype=torch.float64, device=dev)
    assert lib.butd_mlp_mask_stats(P, G * H, G * H, m.data_ptr(), zeros.data_ptr(), ones.data_ptr(), ones.data_ptr(),
                                   zeros[0].data_ptr(), ones.data_ptr(), 0.3, 301 + l * G, H, fa.rng_counter(dev).data_ptr(),
                                   S[0].data_ptr(), S[1].data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    masks.append(m)
def emulated():
    with torch.no_grad():
        for m, st in zip(mlps, state):
            for n, b in m.named_buffers(): b.copy_(st[n])
    x.grad = None
    for m in mlps: m.zero_grad()
    ys = []
    for i, m in enumerate(mlps):
        net = m.net
        t = x.t().unsqueeze(0)                                   # (1, H, P)
        h = torch.relu(net[1](net[0](t))) * masks[0][:, i * H:(i + 1) * H].t().unsqueeze(0)
        h = torch.relu(net[5](net[4](h))) * masks[1][:, i * H:(i + 1) * H].t().unsqueeze(0)
        ys.append(net[8](h)[0].t())
    sum((y * p).sum() for y, p in zip(ys, probes)).backward()
    return [y.detach().clone() for y in ys], [x.grad.clone()] + [p.grad.clone() for m in mlps for p in m.parameters()]
def rel(a, b): return float((a - b).abs().max() / b.abs().max().clamp_min(1e-9))
y_ref, g_ref = emulated()
for fuse in (False, True):
    y, g = run(fuse)
    print("fuse", fuse, "fwd", max(rel(a, b) for a, b in zip(y, y_ref)), "grads", " ".join(f"{rel(a, b):.1e}" for a, b in zip(g, g_ref)))
