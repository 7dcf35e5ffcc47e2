"""round 3: torch's multi-block reduction (vector_norm / sum of a 21 M-element buffer) inside a replayed hipGraph:
is the result always written when the graph is launched behind a still-running graph?"""
import os, sys, torch
dev = torch.device("cuda", 0)
n = 21_400_000
flat = torch.randn(n, device=dev)
a = torch.randn(2048, 2048, device=dev)
want = float(torch.linalg.vector_norm(flat))
res = torch.zeros(4, device=dev)

def busy():
    x = a
    for _ in range(int(os.environ.get("BUSY", "40"))):
        x = torch.tanh(x @ a * 1e-2)
    return x

def reduce_():
    nrm = torch.linalg.vector_norm(flat)
    res[0].copy_(nrm)
    torch.clamp(0.1 / (nrm + 1e-6), max=1.0, out=res[1])
    res[2].copy_(flat.sum())
    res[3].copy_((flat * flat).sum().sqrt())

s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        busy(); reduce_()
    torch.cuda.synchronize()
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        keep = busy()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, pool=g1.pool() if os.environ.get("SHARE_POOL", "1") == "1" else None):
        reduce_()
torch.cuda.synchronize()
iters = 300
for mode in ("sync_between", "free"):
    log = torch.zeros(iters, 4, device=dev)
    for i in range(iters):
        res.fill_(-1.0)
        g1.replay()
        if mode == "sync_between":
            torch.cuda.synchronize()
        g2.replay()
        log[i].copy_(res)
    torch.cuda.synchronize()
    log = log.cpu()
    ref = log[0] if mode == "sync_between" else ref
    bad = ((log - ref).abs() > 1e-3 * ref.abs().clamp_min(1e-6)).any(1)
    print(mode, "want norm", round(want, 3), "first row", [round(float(v), 5) for v in log[0]], "bad replays:", int(bad.sum()), "of", iters)
    if bad.any():
        idx = bad.nonzero().flatten()[:5].tolist()
        print("   examples", [(i, [round(float(v), 5) for v in log[i]]) for i in idx])
