"""two identical models trained side by side (same batches, same dropout counter per step, stock Philox dropouts off):
their losses must stay together; a timing-dependent error in the captured step drives them apart."""
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
batches = [synthetic_batch(args.batch, dev, seed=1184 + 50 * i, n_points=args.points, tokens=args.tokens) for i in range(3)]
def make():
    model = copy.deepcopy(base)
    return GraphedTrainStep(model, FlatAdamW(model), criterion=bench.make_criterion(args), warmup=1), model
(a, ma), (b, mb) = make(), make()
la, lb = [], []
for it in range(int(os.environ.get("STEPS", "45"))):
    k = it % 3
    for s, out in ((a, la), (b, lb)):
        fa.rng_counter(dev).fill_(1000 + it)
        out.append(s(*batches[k], next_inputs=batches[(k + 1) % 3][0]))
        torch.cuda.synchronize()
la, lb = [float(x) for x in la], [float(x) for x in lb]
print("loss a:", [round(x, 3) for x in la[4::5]])
print("loss b:", [round(x, 3) for x in lb[4::5]])
print("max |la - lb| over the run: %.4f" % max(abs(x - y) for x, y in zip(la, lb)))
pa = torch.cat([p.detach().flatten() for p in ma.parameters() if p.requires_grad]); pb = torch.cat([p.detach().flatten() for p in mb.parameters() if p.requires_grad])
d = (pa - pb).abs()
print("parameters: max |d| %.3e, fraction within 1e-5: %.4f" % (float(d.max()), float((d < 1e-5).float().mean())))
