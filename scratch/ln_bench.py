"""butd_add_dropout_layernorm_fwd / _bwd alone at the step's sizes (rows x 288, dropout 0.1, residual): graph replay of
20 launches, us per launch.  BUTD_LN_RPW / BUTD_LN_ABL select variants (read once per process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from butd_detr_amd import _hiplib
from butd_detr_amd.fused_attention import rng_counter
lib = _hiplib.load()
dev = torch.device("cuda", 0)
E_ = 288
out = []
for rows in (640, 2048, 8192, 65536):
    x, res, dy = (torch.randn(rows, E_, device=dev) for _ in range(3))
    gamma, beta = torch.randn(E_, device=dev), torch.randn(E_, device=dev)
    y, dx, dres = (torch.empty(rows, E_, device=dev) for _ in range(3))
    mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    dg, db = torch.zeros(E_, device=dev), torch.zeros(E_, device=dev)
    ctr = rng_counter(dev)
    s = torch.cuda.Stream()
    def fwd():
        lib.butd_add_dropout_layernorm_fwd(rows, E_, x.data_ptr(), res.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5,
                                           y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), 0.1, 7, ctr.data_ptr(), s.cuda_stream)
    nb = lib.butd_layernorm_bwd_blocks(rows)
    part = torch.empty(nb, 2 * E_, device=dev)
    def bwd():      # the product's form: per-workgroup partial sums of dgamma / dbeta, folded by the next grouped launch
        lib.butd_add_dropout_layernorm_bwd_partial(rows, E_, dy.data_ptr(), x.data_ptr(), res.data_ptr(), gamma.data_ptr(),
                                                   mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), dres.data_ptr(),
                                                   part.data_ptr(), 0.1, 7, ctr.data_ptr(), s.cuda_stream)
    with torch.cuda.stream(s):
        fwd(); bwd()
    torch.cuda.synchronize()
    t = {}
    for name, f in (("fwd", fwd), ("bwd", bwd)):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                for _ in range(20): f()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): g.replay()
        b.record(); torch.cuda.synchronize()
        t[name] = a.elapsed_time(b) / 400 * 1e3
    out.append(f"{rows}: fwd {t['fwd']:.1f} bwd {t['bwd']:.1f}")
print(f"RPW={os.environ.get('BUTD_LN_RPW', '-')} ABL={os.environ.get('BUTD_LN_ABL', '-')}  " + "   ".join(out))
