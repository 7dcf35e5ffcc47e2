import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.profiler import profile, ProfilerActivity
from butd_detr_amd.train_step import make_optimizer, synthetic_batch, train_step as eager_step
args = bench.parse()
device = torch.device("cuda", 0)
model, backend = bench.build_model(args, device)
inputs, targets = synthetic_batch(args.batch, device, n_points=args.points, tokens=args.tokens, rank=0)
opt = make_optimizer(model)
for _ in range(2):
    eager_step(model, opt, inputs, targets)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    eager_step(model, opt, inputs, targets)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = [(e.key, e.count, e.self_device_time_total, str(e.input_shapes)[:110]) for e in ka if e.self_device_time_total > 0]
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
print("total device us", tot)
for k, c, t, s in rows[:130]:
    print("%8.1f us  x%-4d %-42s %s" % (t, c, k[:42], s))
