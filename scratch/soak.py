"""soak: 150 graph-replayed steps over rotating batches (prefetch announcements), loss finite, memory flat"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, synthetic_batch
args = bench.parse()
dev = torch.device("cuda", 0)
model, _ = bench.build_model(args, dev)
crit = bench.make_criterion(args)
step = GraphedTrainStep(model, FlatAdamW(model), criterion=crit)
batches = [synthetic_batch(args.batch, dev, seed=1184 + 50 * i, n_points=args.points, tokens=args.tokens) for i in range(3)]
losses, mem0 = [], None
for it in range(150):
    inp, tgt = batches[it % 3]
    nxt = batches[(it + 1) % 3][0]
    loss = step(inp, tgt, next_inputs=nxt)
    if it % 10 == 9:
        torch.cuda.synchronize()
        losses.append(float(loss))
        if mem0 is None: mem0 = torch.cuda.memory_allocated()
assert all(l == l and abs(l) < 1e6 for l in losses), losses
print("losses every 10 steps:", [round(l, 3) for l in losses])
print("memory growth MB:", (torch.cuda.memory_allocated() - mem0) / 1e6, "peak GB:", torch.cuda.max_memory_allocated() / 1e9)
st = crit.set_criterion.matcher.last_status
print("solver status nonzero:", int(st.sum()) if st is not None else None)
