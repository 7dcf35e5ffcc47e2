"""does replaying the SAME hipGraphExec block the host until its previous replay has finished?"""
import torch, time
x = torch.zeros(1 << 26, device="cuda")
def make():
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2): x.add_(1)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(200): x.add_(1)          # ~200 x 0.13 ms = 25 ms of GPU work, 200 nodes
    return g
a, b = make(), make()
torch.cuda.synchronize()
for name, seq in (("A A A A", [a, a, a, a]), ("A B A B", [a, b, a, b])):
    torch.cuda.synchronize()
    ts = []
    t0 = time.perf_counter()
    for g in seq:
        t = time.perf_counter(); g.replay(); ts.append((time.perf_counter() - t) * 1e3)
    host = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) * 1e3
    print(f"{name}: host time per replay {[round(v, 2) for v in ts]} ms, host total {host:.1f} ms, GPU done after {total:.1f} ms")
