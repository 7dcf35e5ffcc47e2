"""soak (90 graphed steps, 3 rotating batches) with GraphedTrainStep options from the environment"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, synthetic_batch
args = bench.parse()
dev = torch.device("cuda", 0)
if os.environ.get("SPLIT") == "1":
    import torch.distributed as dist
    os.environ["BUTD_FORCE_COLLECTIVE"] = "1"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group(os.environ.get("BACKEND", "nccl"), init_method="env://", rank=0, world_size=1)
model, _ = bench.build_model(args, dev)
crit = bench.make_criterion(args)
if os.environ.get("STOCK_DROPOUT", "1") == "0":
    model.text_encoder.eval()
    for m in model.text_projector.modules():
        if isinstance(m, torch.nn.Dropout): m.p = 0.0
kw = dict(prefetch_sampling=os.environ.get("PS", "1") == "1", prefetch_text=os.environ.get("PT", "1") == "1",
          zero_arena=os.environ.get("ZA", "1") == "1", overlap_exchange=os.environ.get("OVERLAP", "0") == "1")
eps = float(os.environ.get("LOSS_EPS", "0"))
if eps:
    inner = crit
    class Scaled:
        def prepare(self, t): return inner.prepare(t)
        def __call__(self, ep, t): return inner(ep, t) * (1.0 + eps)
    crit = Scaled()
step = GraphedTrainStep(model, FlatAdamW(model), criterion=crit, **kw)
batches = [synthetic_batch(args.batch, dev, seed=1184 + 50 * i, n_points=args.points, tokens=args.tokens) for i in range(3)]
losses = []
for it in range(int(os.environ.get("STEPS", "90"))):
    inp, tgt = batches[it % 3]
    loss = step(inp, tgt, next_inputs=batches[(it + 1) % 3][0])
    if it == 0 and os.environ.get("EMPTY_CACHE") == "1":
        torch.cuda.synchronize()
        before = torch.cuda.memory_reserved()
        torch.cuda.empty_cache()                     # cached free blocks of the default pool go back to the driver:
        print("empty_cache released MB:", (before - torch.cuda.memory_reserved()) / 1e6)   # a baked-in stale address would fault
    if it % 10 == 9: losses.append(round(float(loss), 2))
print(kw, {"BUTD_AB": os.environ.get("BUTD_AB")}, losses)
