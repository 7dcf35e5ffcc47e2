"""eager soak with the packed optimizer (FlatAdamW used eagerly, clip 0.1 as the graph step): 60 steps, 3 batches"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from butd_detr_amd.train_step import FlatAdamW, synthetic_batch
args = bench.parse()
dev = torch.device("cuda", 0)
model, _ = bench.build_model(args, dev)
crit = bench.make_criterion(args)
if os.environ.get("STOCK_DROPOUT", "1") == "0":
    model.text_encoder.eval()
    for m in model.text_projector.modules():
        if isinstance(m, torch.nn.Dropout): m.p = 0.0
opt = FlatAdamW(model)
batches = [synthetic_batch(args.batch, dev, seed=1184 + 50 * i, n_points=args.points, tokens=args.tokens) for i in range(3)]
losses = []
for it in range(int(os.environ.get("STEPS", "60"))):
    inp, tgt = batches[it % 3]
    loss = crit(model(inp), crit.prepare(tgt))
    opt.zero_grad()
    loss.backward()
    opt.collect_grads()
    opt.clip_(0.1, grad_div=1.0)
    opt.step(packed=True)
    if it % 10 == 9: losses.append(round(float(loss), 2))
print("eager+FlatAdamW", losses)
