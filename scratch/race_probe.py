"""two identical models, the same batch, the same dropout counter, one graph replay each (lr = 0): the packed gradients
must agree to the rounding of the atomics.  A race inside the captured step shows as a larger difference."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from butd_detr_amd import fused_attention as fa
from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, synthetic_batch
args = bench.parse()
dev = torch.device("cuda", 0)
base, _ = bench.build_model(args, dev)
base.text_encoder.eval()                 # stock Philox dropouts advance per replay: off (the counter-hash ones stay on)
for m in base.text_projector.modules():
    if isinstance(m, torch.nn.Dropout): m.p = 0.0
batches = [synthetic_batch(args.batch, dev, seed=1184 + 50 * i, n_points=args.points, tokens=args.tokens) for i in range(2)]
def make():
    model = copy.deepcopy(base)
    opt = FlatAdamW(model, lr=0.0, lr_backbone=0.0, text_encoder_lr=0.0, weight_decay=0.0)
    return GraphedTrainStep(model, opt, criterion=bench.make_criterion(args), warmup=1,
                            prefetch_text=os.environ.get("PT", "1") == "1", prefetch_sampling=os.environ.get("PS", "1") == "1"), opt
(a, oa), (b, ob) = make(), make()
for s in (a, b): s(*batches[0], next_inputs=batches[1][0])
torch.cuda.synchronize()
worst = 0.0
for trial in range(int(os.environ.get("TRIALS", "8"))):
    gs = []
    for s, o in ((a, oa), (b, ob)):
        fa.rng_counter(dev).fill_(1000 + trial)
        k = trial % 2
        s(*batches[k], next_inputs=batches[1 - k][0])
        torch.cuda.synchronize()
        gs.append(o.flat_g.clone())
    d = float((gs[0] - gs[1]).abs().max() / gs[0].abs().max())
    nz = int(((gs[0] - gs[1]).abs() > 1e-4 * gs[0].abs().max()).sum())
    worst = max(worst, d)
    print(f"trial {trial}: max |dg| / max |g| = {d:.3e}   elements off by > 1e-4 of max: {nz}")
print("worst", worst)
