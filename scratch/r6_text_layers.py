"""round 6: how does the step respond to a SHORTER language-model branch?  The frozen tower truncated to N of its 12 layers
(timing experiment: the features are then not the model's): python r6_text_layers.py N"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, torch, bench
from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, HungarianCriterion, synthetic_batch
dev = torch.device("cuda", 0)
args = argparse.Namespace(backend="auto", queries=256, points=50000, tokens=80, encoder_layers=3)
batches = [synthetic_batch(8, dev, seed=1184 + 50 * i, n_points=50000, tokens=80) for i in range(4)]
n = int(sys.argv[1])
model, _ = bench.build_model(args, dev)
enc = model.text_encoder.encoder
enc.layer = torch.nn.ModuleList(list(enc.layer)[:n])
step = GraphedTrainStep(model, FlatAdamW(model), criterion=HungarianCriterion())
for it in range(6):
    step(batches[it % 4][0], batches[it % 4][1], next_inputs=batches[(it + 1) % 4][0])
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for it in range(60):
    step(batches[it % 4][0], batches[it % 4][1], next_inputs=batches[(it + 1) % 4][0])
b.record(); torch.cuda.synchronize()
print(f"language model with {n} of 12 layers: {a.elapsed_time(b) / 60:.3f} ms / step")
