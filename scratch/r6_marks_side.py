"""round 6: in-situ regions of the captured step (StepMarks) with the prefetch branches on / off / replaced by one tiny launch:
WHERE does the cost of a branch land?  python r6_marks_side.py {base|none|dummy1}"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, torch, bench
from butd_detr_amd import step_regions
from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, HungarianCriterion, synthetic_batch
dev = torch.device("cuda", 0)
args = argparse.Namespace(backend="auto", queries=256, points=50000, tokens=80, encoder_layers=3)
batches = [synthetic_batch(8, dev, seed=1184 + 50 * i, n_points=50000, tokens=80) for i in range(4)]
mode = sys.argv[1]
model, _ = bench.build_model(args, dev)
step = GraphedTrainStep(model, FlatAdamW(model), criterion=HungarianCriterion())
o_s, o_t = step._sample_into_next, step._encode_text_into_next
cap = torch.cuda.is_current_stream_capturing
cell = torch.zeros(64, device=dev)
if mode in ("none", "dummy1"):
    step._sample_into_next = lambda: None if cap() else o_s()
    step._encode_text_into_next = (lambda: None if cap() else o_t()) if mode == "none" else (lambda: cell.add_(1.0) if cap() else o_t())
marks = step_regions.StepMarks(model, step, dev).install()
for it in range(6):
    step(batches[it % 4][0], batches[it % 4][1], next_inputs=batches[(it + 1) % 4][0])
res = marks.measure(batches, replays=40, skip=4)
iv = res["intervals"]
inside = sum(u for _, _, u, _ in iv) / 1e3
print(f"{mode}: {res['ms_per_step']:.3f} ms per step; first mark -> last mark {inside:.3f} ms; outside the marks {res['ms_per_step'] - inside:.3f} ms")
for k, v in res["regions"].items():
    if isinstance(v, dict) and "ms" in v and "tflops" not in v:
        print(f"   {k:24s} {v['ms']:.3f}")
