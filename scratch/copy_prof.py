"""who launches the copy / fill / add kernels of one eager step (python call sites via with_stack)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.profiler import profile, ProfilerActivity
from butd_detr_amd.train_step import make_optimizer, synthetic_batch, train_step as eager_step
args = bench.parse()
device = torch.device("cuda", 0)
model, backend = bench.build_model(args, device)
inputs, targets = synthetic_batch(args.batch, device, n_points=args.points, tokens=args.tokens, rank=0)
crit = bench.make_criterion(args)
from butd_detr_amd.fused_attention import ZeroArena
opt = make_optimizer(model)
for _ in range(2):
    eager_step(model, opt, inputs, targets, criterion=crit)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    end_points = model(inputs)
    loss = crit(end_points, crit.prepare(targets))
    loss.backward()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::add", "aten::add_", "aten::cat", "aten::zero_", "aten::mul", "aten::sum") and e.device_time_total > 0:
        site = "?"
        for fr in e.stack or []:
            if "/butd_detr_amd/" in fr or "bench.py" in fr:
                site = fr.split("/butd_detr_amd/")[-1][:70]
                break
        key = (e.name, str(e.input_shapes)[:60], site)
        agg[key][0] += 1
        agg[key][1] += e.device_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print("total us", sum(v[1] for v in agg.values()))
for (name, shp, site), (n, t) in rows[:60]:
    print(f"{t:8.1f} us x{n:<4d} {name:12s} {shp:60s} {site}")
