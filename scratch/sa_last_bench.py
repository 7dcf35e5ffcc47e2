"""round 4: butd_sa_last_bwd alone at the bench's SA1 / SA2 sizes (graph replay), with the ablation hook."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from butd_detr_amd import _hiplib
lib = _hiplib.load()
lib.butd_sa_last_bwd_set_ablation.argtypes = [ctypes.c_int]
dev = torch.device("cuda", 0)


def bench(B, np_, ns, C2, C3, abl):
    G, P = B * np_, B * np_ * ns
    torch.manual_seed(0)
    Z2 = torch.randn(P, C2, device=dev)
    aff2 = [torch.rand(C2, device=dev) + 0.5 for _ in range(4)]
    aff3 = [torch.rand(C3, device=dev) + 0.5 for _ in range(4)]
    W3 = torch.randn(C3, C2, device=dev) / 8
    d_out, zsel = torch.randn(G, C3, device=dev), torch.randn(G, C3, device=dev)
    asel = torch.randint(0, ns, (G, C3), device=dev, dtype=torch.uint8)
    S3 = torch.randn(2, C3, device=dev, dtype=torch.float64)
    dH2, dW3 = torch.empty(P, C2, device=dev), torch.empty(C3, C2, device=dev)
    S2 = torch.empty(2, C2, device=dev, dtype=torch.float64)
    nf, nd = ctypes.c_long(0), ctypes.c_long(0)
    lib.butd_sa_last_bwd_scratch(P, C2, C3, ctypes.byref(nf), ctypes.byref(nd))
    wf, wd = torch.empty(nf.value, device=dev), torch.empty(nd.value, device=dev, dtype=torch.float64)
    lib.butd_sa_last_bwd_set_ablation(abl)
    st = torch.cuda.Stream()
    def run():
        e = lib.butd_sa_last_bwd(B, np_, ns, C2, C3, Z2.data_ptr(), aff2[0].data_ptr(), aff2[1].data_ptr(), aff2[2].data_ptr(),
                                 aff2[3].data_ptr(), W3.data_ptr(), d_out.data_ptr(), zsel.data_ptr(), asel.data_ptr(),
                                 aff3[0].data_ptr(), aff3[1].data_ptr(), aff3[2].data_ptr(), aff3[3].data_ptr(),
                                 S3[0].data_ptr(), S3[1].data_ptr(), dH2.data_ptr(), dW3.data_ptr(), S2[0].data_ptr(),
                                 S2[1].data_ptr(), wf.data_ptr(), wd.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert e == 0, e
    with torch.cuda.stream(st):
        run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            run()
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record(); torch.cuda.synchronize()
    lib.butd_sa_last_bwd_set_ablation(0)
    return e0.elapsed_time(e1) / 20 * 1e3


for name, cfg in (("SA1", (8, 2048, 64, 64, 128)), ("SA2", (8, 1024, 32, 128, 256)), ("SA3", (8, 512, 16, 128, 256)), ("SA4", (8, 256, 16, 128, 256))):
    print(name, "  ".join(f"abl={a}: {bench(*cfg, a):7.1f} us" for a in (0,)))


def bench_fwd(B, np_, ns, C2, C3):
    G, P = B * np_, B * np_ * ns
    torch.manual_seed(0)
    Z2 = torch.randn(P, C2, device=dev)
    sc, sh = torch.rand(C2, device=dev) + 0.5, torch.randn(C2, device=dev) * 0.1
    W3 = torch.randn(C3, C2, device=dev) / 8
    st8 = torch.zeros(2, C3, device=dev, dtype=torch.float64)
    zmax, zmin = torch.empty(G, C3, device=dev), torch.empty(G, C3, device=dev)
    amax, amin = torch.empty(G, C3, device=dev, dtype=torch.uint8), torch.empty(G, C3, device=dev, dtype=torch.uint8)
    sched = torch.zeros(2, dtype=torch.int32, device=dev)
    st = torch.cuda.Stream()
    def run():
        e = lib.butd_sa_last_fwd(B, np_, ns, C2, C3, Z2.data_ptr(), sc.data_ptr(), sh.data_ptr(), W3.data_ptr(), st8[0].data_ptr(),
                                 st8[1].data_ptr(), zmax.data_ptr(), zmin.data_ptr(), amax.data_ptr(), amin.data_ptr(),
                                 sched.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert e == 0, e
    with torch.cuda.stream(st):
        run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            run()
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


for name, cfg in (("SA1", (8, 2048, 64, 64, 128)), ("SA2", (8, 1024, 32, 128, 256)), ("SA3", (8, 512, 16, 128, 256)), ("SA4", (8, 256, 16, 128, 256))):
    P = cfg[0] * cfg[1] * cfg[2]
    t = bench_fwd(*cfg)
    print(f"fwd {name}: {t:7.1f} us   {2 * P * cfg[3] * cfg[4] / t / 1e6:6.1f} TF   Z2 read {P * cfg[3] * 4 / t / 1e6:6.2f} TB/s")
