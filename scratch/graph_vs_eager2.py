"""like graph_vs_eager.py but TRAINING (reference learning rates): before every replay the eager model takes the graph
model's current parameters and buffers, both evaluate the same batch with the same dropout counter; the replay's
gradients must equal the eager ones at every step."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from butd_detr_amd import fused_attention as fa
from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, synthetic_batch
args = bench.parse()
dev = torch.device("cuda", 0)
if os.environ.get("SPLIT") == "1":
    import torch.distributed as dist
    os.environ["BUTD_FORCE_COLLECTIVE"] = "1"
    dist.init_process_group("gloo", init_method="file:///tmp/gve2_init_%d" % os.getpid(), rank=0, world_size=1)
base, _ = bench.build_model(args, dev)
base.text_encoder.eval()
for m in base.text_projector.modules():
    if isinstance(m, torch.nn.Dropout): m.p = 0.0
batches = [synthetic_batch(args.batch, dev, seed=1184 + 50 * i, n_points=args.points, tokens=args.tokens) for i in range(3)]
crit = bench.make_criterion(args)
me, mg = copy.deepcopy(base), copy.deepcopy(base)
opt = FlatAdamW(mg)
step = GraphedTrainStep(mg, opt, criterion=bench.make_criterion(args), warmup=1, overlap_exchange=os.environ.get('SPLIT') == '1',
                        prefetch_text=os.environ.get('PT', '1') == '1', prefetch_sampling=os.environ.get('PS', '1') == '1')
print('split', step.split, 'prefetch_text', step.prefetch_text, 'prefetch_sampling', step.prefetch_sampling)
step(*batches[0], next_inputs=batches[1][0]); torch.cuda.synchronize()
names = [n for n, p in mg.named_parameters() if p.requires_grad]
for it in range(int(os.environ.get("STEPS", "12"))):
    k = (it + 1) % 3
    with torch.no_grad():
        for (n, pe), (_, pg) in zip(me.named_parameters(), mg.named_parameters()): pe.copy_(pg)
        for (n, be), (_, bg) in zip(me.named_buffers(), mg.named_buffers()): be.copy_(bg)
    c = 700 + it
    for p in me.parameters(): p.grad = None
    fa.rng_counter(dev).fill_(c - 1)
    le = crit(me(batches[k][0]), crit.prepare(batches[k][1])); le.backward()
    torch.cuda.synchronize()
    fa.rng_counter(dev).fill_(c - 1)
    lg = step(*batches[k], next_inputs=batches[(k + 1) % 3][0]); torch.cuda.synchronize()
    rows = []
    off = 0
    for p in opt.params:
        n = p.numel(); pad = (n + 3) // 4 * 4
        rows.append(opt.flat_g[off:off + n]); off += pad
    worst = (0.0, "")
    ge = {n: p.grad for n, p in me.named_parameters() if p.requires_grad}
    pe_list = [p for p in me.parameters() if p.requires_grad]
    # opt.params order != named order: match by identity position through the graph model
    gid = {id(p): n for n, p in mg.named_parameters()}
    for p, g in zip(opt.params, rows):
        n = gid[id(p)]
        e = ge.get(n)
        if e is None: continue
        sc = float(e.abs().max())
        if sc > 1e-6:
            r = float((e.flatten() - g).abs().max()) / sc
            if r > worst[0]: worst = (r, n)
    print(f"step {it}: loss eager {float(le):.4f} graph {float(lg):.4f}   worst relative gradient difference {worst[0]:.3e} ({worst[1]})")
