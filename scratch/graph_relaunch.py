"""is re-launching a hipGraphExec while its previous launch is still executing safe?  (values, not only host time)"""
import torch, time
def make(n_nodes, fork):
    x = torch.zeros(1 << 24, device="cuda"); y = torch.zeros(1 << 24, device="cuda")
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); side = torch.cuda.Stream()
    with torch.cuda.stream(s):
        x.add_(1); y.add_(1)
        torch.cuda.synchronize()
        x.zero_(); y.zero_()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for i in range(n_nodes):
                x.add_(1)
                if fork and i % 8 == 0:
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        y.add_(x)                 # reads x on a forked branch
                    torch.cuda.current_stream().wait_stream(side)
    return g, x, y
for n_nodes, fork in ((200, False), (1400, False), (1400, True)):
    g, x, y = make(n_nodes, fork)
    torch.cuda.synchronize()
    reps = 6
    t0 = time.perf_counter(); ts = []
    for _ in range(reps):
        t = time.perf_counter(); g.replay(); ts.append(round((time.perf_counter() - t) * 1e3, 2))
    torch.cuda.synchronize()
    want_x = float(n_nodes * reps)
    ok_x = bool((x == want_x).all())
    msg = f"nodes {n_nodes} fork {fork}: host ms per replay {ts}; x == {want_x}: {ok_x}"
    if fork:
        # y accumulates x at i = 0, 8, 16, ...: per replay r (0-based): sum over those i of (r * n + i + 1)
        idx = list(range(0, n_nodes, 8))
        want_y = float(sum(r * n_nodes + i + 1 for r in range(reps) for i in idx))
        msg += f"; y == {want_y}: {bool((y == want_y).all())} (got {float(y[0])})"
    print(msg)
