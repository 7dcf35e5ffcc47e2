"""round 3: are MEMSET / MEMCPY nodes of a hipGraph ordered after the kernel nodes that precede them in the same
graph when the graph is launched behind a still-running graph?  (ROCm 7.2, MI355X)"""
import ctypes, os, sys, torch
dev = torch.device("cuda", 0)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
a = torch.randn(2048, 2048, device=dev)
buf = torch.zeros(1 << 20, device=dev)            # 4 MB
small = torch.zeros(64, device=dev)               # 256 B (semaphore-sized)
A = torch.zeros(1 << 20, device=dev); B = torch.zeros(1 << 20, device=dev)
out = torch.zeros(4, device=dev)

def busy(k=40):
    x = a
    for _ in range(k):
        x = torch.tanh(x @ a * 1e-2)
    return x

def body():
    st = torch.cuda.current_stream().cuda_stream
    k = busy(4)                                   # some kernels first
    buf.fill_(1.0); small.fill_(1.0)              # kernel nodes
    hip.hipMemsetAsync(buf.data_ptr(), 0, buf.numel() * 4, st)        # MEMSET node (large)
    hip.hipMemsetAsync(small.data_ptr(), 0, small.numel() * 4, st)    # MEMSET node (small)
    out[0].copy_(buf[:4096].sum()); out[1].copy_(small.sum())         # expected 0, 0
    A.add_(1.0)                                   # kernel node
    B.copy_(A)                                    # MEMCPY node (D2D)
    out[2].copy_(B[:1024].mean())                 # expected: replay index + warmups
    return k

s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2):
        busy(); body()
    torch.cuda.synchronize()
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        keep1 = busy()
    g2 = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g2):
        keep2 = body()
torch.cuda.synchronize()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from butd_detr_amd import graph_audit
print("g2 nodes:", dict(graph_audit.inventory(g2)))
if os.environ.get("REWRITE") == "1":
    print("memset nodes rewritten into kernel nodes:", graph_audit.make_safe(g2), "->", dict(graph_audit.inventory(g2)))
iters = 200
for mode in ("sync_between", "behind_g1", "behind_itself"):
    A.zero_(); torch.cuda.synchronize()
    log = torch.zeros(iters, 4, device=dev)
    for i in range(iters):
        if mode != "behind_itself":
            g1.replay()
        if mode == "sync_between":
            torch.cuda.synchronize()
        g2.replay()
        log[i].copy_(out)
    torch.cuda.synchronize()
    log = log.cpu()
    bad_big = int((log[:, 0] != 0).sum()); bad_small = int((log[:, 1] != 0).sum())
    want = torch.arange(1, iters + 1, dtype=torch.float32)
    bad_cpy = int((log[:, 2] != want).sum())
    print(f"{mode}: memset(4MB) wrong {bad_big}/{iters}, memset(256B) wrong {bad_small}/{iters}, memcpy wrong {bad_cpy}/{iters}",
          "e.g.", log[:3].tolist())
