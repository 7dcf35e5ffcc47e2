"""which aten ops of one eager training step call hipMemsetAsync (they become MEMSET nodes of the captured graph)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = sys.argv[:1]
import torch, bench
from torch.profiler import profile, ProfilerActivity
from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, synthetic_batch
args = bench.parse()
dev = torch.device("cuda", 0)
model, _ = bench.build_model(args, dev)
crit = bench.make_criterion(args)
opt = FlatAdamW(model)
step = GraphedTrainStep(model, opt, criterion=crit, prefetch_sampling=False, prefetch_text=False)
inp, tgt = synthetic_batch(args.batch, dev, n_points=args.points, tokens=args.tokens)
tgt = crit.prepare(tgt)
from transformers import BatchEncoding
tok = step._tokenize(inp)
class S: pass
s = step._slot = S(); s.inputs, s.targets, s.tok, s.cut = inp, tgt, tok, None
for _ in range(2): step._one_eager_step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step._one_eager_step()
    torch.cuda.synchronize()
evs = prof.events()
ops = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU and not e.name.startswith("hip")]
sets = [e for e in evs if "emset" in e.name]
print("memset-like events:", collections.Counter(e.name for e in sets))
count = collections.Counter()
for m in sets:
    if m.device_type != torch.autograd.DeviceType.CPU: continue
    t0, t1 = m.time_range.start, m.time_range.end
    enc = [o for o in ops if o.time_range.start <= t0 and o.time_range.end >= t1]
    enc.sort(key=lambda o: o.time_range.end - o.time_range.start)
    chain = " < ".join(f"{o.name}{list(o.input_shapes)[:2] if o.input_shapes else ''}" for o in enc[:3])
    count[chain] += 1
for k, v in count.most_common():
    print(v, k)
