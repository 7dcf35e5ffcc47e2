"""flat gradients of ONE graphed step (lr = 0, dropout 0) under different code paths: where do they differ?"""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, synthetic_batch
args = bench.parse()
dev = torch.device("cuda", 0)
base, _ = bench.build_model(args, dev)
for m in base.modules():
    if isinstance(m, torch.nn.Dropout): m.p = 0.0
    if hasattr(m, "dropout") and isinstance(getattr(m, "dropout"), float): m.dropout = 0.0
batch = synthetic_batch(args.batch, dev, seed=1184, n_points=args.points, tokens=args.tokens)
def run(env, eager=False):
    for k, v in env.items(): os.environ[k] = v
    model = copy.deepcopy(base)
    crit = bench.make_criterion(args)
    opt = FlatAdamW(model, lr=0.0, lr_backbone=0.0, text_encoder_lr=0.0, weight_decay=0.0)
    step = GraphedTrainStep(model, opt, criterion=crit, warmup=1)
    for _ in range(3): loss = step(*batch)
    torch.cuda.synchronize()
    return float(loss), opt.flat_g.clone(), [(n, p.numel()) for n, p in model.named_parameters() if p.requires_grad], opt
configs = {"both on": {"BUTD_FAN_OUT": "1", "BUTD_PROJ_CHAIN": "1"}, "both off": {"BUTD_FAN_OUT": "0", "BUTD_PROJ_CHAIN": "0"},
           "chain only": {"BUTD_FAN_OUT": "0", "BUTD_PROJ_CHAIN": "1"}, "both off again": {"BUTD_FAN_OUT": "0", "BUTD_PROJ_CHAIN": "0"}}
res = {k: run(v) for k, v in configs.items()}
ref = res["both on"]
names = {id(p): n for n, p in copy.deepcopy(base).named_parameters()}
for k, (loss, g, _, opt) in res.items():
    d = (g - ref[1]).abs()
    print(f"{k:16s} loss {loss:.5f}  |g| max {float(g.abs().max()):.4e}  max |dg| {float(d.max()):.4e}  rel {float(d.max() / ref[1].abs().max()):.3e}  norm ratio {float(g.norm() / ref[1].norm()):.5f}")
# which parameters differ most between "both off" and "both on"
opt = res["both off"][3]
g_off, g_on = res["both off"][1], ref[1]
off = 0
rows = []
for p in opt.params:
    n = p.numel(); pad = (n + 3) // 4 * 4
    a, b = g_off[off:off + n], g_on[off:off + n]
    rows.append((float((a - b).abs().max() / (b.abs().max() + 1e-12)), float(b.abs().max()), n, off)); off += pad
pn = [n for n, p in res["both off"][3].__dict__.get("named", [])] if False else None
model_names = [n for n, p in base.named_parameters() if p.requires_grad]
print("params with the largest relative gradient difference (both off vs both on):")
order = sorted(range(len(rows)), key=lambda i: -rows[i][0])[:12]
for i in order: print(f"  #{i:4d} rel {rows[i][0]:.3e}  |g|max {rows[i][1]:.3e}  numel {rows[i][2]}")
