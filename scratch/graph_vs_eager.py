"""gradients of a hipGraph replay vs the eager evaluation of the SAME model, batch and dropout counter (stock Philox
dropouts off, lr = 0): a dependency missing from the captured graph shows as a systematic difference."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from butd_detr_amd import fused_attention as fa
from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, synthetic_batch
args = bench.parse()
dev = torch.device("cuda", 0)
base, _ = bench.build_model(args, dev)
base.text_encoder.eval()
for m in base.text_projector.modules():
    if isinstance(m, torch.nn.Dropout): m.p = 0.0
batch = synthetic_batch(args.batch, dev, seed=1184, n_points=args.points, tokens=args.tokens)
crit = bench.make_criterion(args)
# eager
me = copy.deepcopy(base)
tg = crit.prepare(batch[1])
def eager_grads(counter):
    for p in me.parameters(): p.grad = None
    fa.rng_counter(dev).fill_(counter - 1)          # forward bumps it by one
    crit(me(batch[0]), tg).backward()
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in me.named_parameters() if p.grad is not None}
# graph
mg = copy.deepcopy(base)
opt = FlatAdamW(mg, lr=0.0, lr_backbone=0.0, text_encoder_lr=0.0, weight_decay=0.0)
step = GraphedTrainStep(mg, opt, criterion=bench.make_criterion(args), warmup=1)
step(*batch); torch.cuda.synchronize()
worst = []
for trial in range(3):
    c = 500 + trial
    ge = eager_grads(c)
    fa.rng_counter(dev).fill_(c - 1)
    step(*batch); torch.cuda.synchronize()
    gg = {n: p.grad.clone() for n, p in mg.named_parameters() if p.grad is not None}
    rows = []
    for n in ge:
        if n in gg:
            sc = float(ge[n].abs().max())
            if sc > 1e-6: rows.append((float((ge[n] - gg[n]).abs().max()) / sc, n, sc))
    rows.sort(reverse=True)
    print(f"trial {trial}: worst relative gradient differences graph vs eager:")
    for r in rows[:6]: print(f"   {r[0]:.3e}  {r[1]}  (|g|max {r[2]:.3e})")
    worst.append(rows[0][0])
print("worst:", worst)
