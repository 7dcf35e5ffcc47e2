"""soak in EAGER mode (torch AdamW, no graph): 90 steps over 3 rotating batches"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from butd_detr_amd.train_step import make_optimizer, synthetic_batch, train_step
args = bench.parse()
dev = torch.device("cuda", 0)
model, _ = bench.build_model(args, dev)
crit = bench.make_criterion(args)
opt = make_optimizer(model)
batches = [synthetic_batch(args.batch, dev, seed=1184 + 50 * i, n_points=args.points, tokens=args.tokens) for i in range(3)]
losses = []
for it in range(90):
    inp, tgt = batches[it % 3]
    loss = train_step(model, opt, inp, tgt, criterion=crit)
    if it % 10 == 9: losses.append(round(float(loss), 3))
print("eager losses every 10 steps:", losses)
