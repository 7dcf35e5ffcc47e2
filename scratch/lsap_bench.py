"""butd_hungarian_match alone at the step's size (56 problems = 7 prefixes x 8 scenes, 256 queries, 132 slots of which
1..16 take part) and at the detection-split size (66..132 valid slots): graph replay of 20 launches, us per launch.
Run once as is and once with BUTD_LSAP_NO_STAGE=1 (cost rows read from global memory: the round-3 kernel's data path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from butd_detr_amd import losses as L
rng = np.random.default_rng(0)
def case(lo, hi, P=56, G=132, Q=256):
    cost = torch.from_numpy(rng.standard_normal((P, G, Q)).astype(np.float32) * 3).cuda()
    valid = np.zeros((P, G), dtype=bool)
    for p in range(P):
        valid[p, :rng.integers(lo, hi + 1)] = True
    return cost, torch.from_numpy(valid).cuda()
for name, (lo, hi) in {"step (1..16 targets)": (1, 16), "detection split (66..132)": (66, 132)}.items():
    cost, valid = case(lo, hi)
    L.hungarian_match(cost, valid); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(20):
                L.hungarian_match(cost, valid)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): g.replay()
    b.record(); torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b) / 200 * 1e3:.1f} us per launch (staging {'off' if os.environ.get('BUTD_LSAP_NO_STAGE') else 'on'})")
