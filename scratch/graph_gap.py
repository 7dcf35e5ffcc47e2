"""per-node cost of a hipGraph replay: N tiny dependent kernels"""
import torch, time
x = torch.zeros(64, device="cuda")
big = torch.zeros(8 * 1024 * 288, device="cuda")
for name, t in (("tiny (64 elems)", x), ("9.4 MB add", big)):
    for N in (500, 2000):
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3): t.add_(1)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for _ in range(N): t.add_(1)
        torch.cuda.synchronize()
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"{name:16s} N={N}: {dt*1e3:7.3f} ms per replay = {dt/N*1e6:6.2f} us per node")
