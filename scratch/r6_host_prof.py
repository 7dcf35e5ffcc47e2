"""round 6: where does the host spend a GraphedTrainStep call?  (free-running replays; wall-clock per section, no profiler)"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, torch, bench
from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, HungarianCriterion, synthetic_batch
dev = torch.device("cuda", 0)
args = argparse.Namespace(backend="auto", queries=256, points=50000, tokens=80, encoder_layers=3)
batches = [synthetic_batch(8, dev, seed=1184 + 50 * i, n_points=50000, tokens=80) for i in range(4)]
model, _ = bench.build_model(args, dev)
crit = HungarianCriterion()
opt = FlatAdamW(model)
step = GraphedTrainStep(model, opt, criterion=crit)
acc = collections.defaultdict(float)
if os.environ.get("NONE") == "1":        # both prefetch branches captured as nothing (timing experiment)
    o_s, o_t = step._sample_into_next, step._encode_text_into_next
    cap = torch.cuda.is_current_stream_capturing
    step._sample_into_next = lambda: None if cap() else o_s()
    step._encode_text_into_next = lambda: None if cap() else o_t()


def wrap(obj, name, tag):
    orig = getattr(obj, name)
    def f(*a, **k):
        t = time.perf_counter()
        try:
            return orig(*a, **k)
        finally:
            acc[tag] += time.perf_counter() - t
    setattr(obj, name, f)


for it in range(6):
    step(batches[it % 4][0], batches[it % 4][1], next_inputs=batches[(it + 1) % 4][0])
torch.cuda.synchronize()
wrap(step, "_tokenize", "tokenize"); wrap(crit, "prepare", "criterion.prepare"); wrap(step, "_copy_in", "copy_in")
wrap(opt, "sync_hyper", "sync_hyper")
sl = step._slot
for nm in ("g_fwd_bwd", "g_update"):
    g = getattr(sl, nm)
    orig = g.replay
    class _W:
        def __init__(self, g, tag): self.g, self.tag = g, tag
        def replay(self):
            t = time.perf_counter(); self.g.replay(); acc[self.tag] += time.perf_counter() - t
        def __getattr__(self, k): return getattr(self.g, k)
    setattr(sl, nm, _W(g, "replay " + nm))
wrap(step, "_exchange_whole", "exchange")
reps = 40
t0 = time.perf_counter()
for it in range(reps):
    step(batches[it % 4][0], batches[it % 4][1], next_inputs=batches[(it + 1) % 4][0])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host per call {1e3 * (t1 - t0) / reps:.2f} ms, drained {1e3 * (t2 - t0) / reps:.2f} ms per step")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:20s} {1e3 * v / reps:8.3f} ms per step")
