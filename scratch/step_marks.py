"""round 4: profiler-free in-situ timeline of the captured training step (bench configuration).

Region boundaries are one-wave kernels that write the device wall clock (butd_timeline_mark, s_memrealtime) when the
MAIN stream reaches them -- forward marks after a module's forward, backward marks when autograd passes the module's
output gradient (identity Function).  The marks are captured into the step's hipGraph; after N replays the per-region
durations are averaged.  Each mark costs one launch (~3.5 us) on the main queue: ~60 marks = 0.2 ms of inflation.

    python scratch/step_marks.py [replays]
"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import torch
import bench
from butd_detr_amd import _hiplib
from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, HungarianCriterion, synthetic_batch

lib = _hiplib.load()
dev = torch.device("cuda", 0)
MAXS = 512
slots = torch.zeros(MAXS, dtype=torch.int64, device=dev)
names = []


def mark(name):
    if len(names) >= MAXS:
        return
    names.append(name)
    lib.butd_timeline_mark(slots.data_ptr(), len(names) - 1, torch.cuda.current_stream(dev).cuda_stream)


class _BwdMark(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, name):
        ctx.name = name
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        mark("bwd<" + ctx.name)     # autograd is about to run this module's backward (its output gradient is ready)
        return g, None


def first_tensor_map(out, fn):
    if torch.is_tensor(out):
        return fn(out) if out.requires_grad else out
    if isinstance(out, tuple):
        done, res = False, []
        for o in out:
            if not done and torch.is_tensor(o) and o.requires_grad:
                res.append(fn(o)); done = True
            else:
                res.append(o)
        return tuple(res)
    return out


def hook(mod, name):
    orig = mod.forward

    def f(*a, **k):
        out = orig(*a, **k)
        if torch.cuda.current_stream(dev) != main_stream[0]:
            return out
        mark("fwd>" + name)
        return first_tensor_map(out, lambda t: _BwdMark.apply(t, name))
    mod.forward = f


main_stream = [None]
args = argparse.Namespace(backend="auto", queries=256, points=50000, tokens=80, encoder_layers=3)
model, backend = bench.build_model(args, dev)
m = model
for n in ("sa1", "sa2", "sa3", "sa4", "fp1", "fp2"):
    hook(getattr(m.backbone_net, n), n)
for i, l in enumerate(m.cross_encoder.layers):
    hook(l, f"enc{i}")
for i, l in enumerate(m.decoder):
    hook(l, f"dec{i}")
for i, l in enumerate(m.prediction_heads):
    hook(l, f"head{i}")
hook(m.proposal_head, "proposal_head")
hook(m.points_obj_cls, "objcls")
hook(m.pos_embed, "pos_embed")
hook(m.text_projector, "text_proj")
hook(m.gsample_module, "gsample")
hook(m.decoder_query_proj, "query_proj")
hook(m.contrastive_align_projection_image, "proj_img")
hook(m.contrastive_align_projection_text, "proj_txt")
if os.environ.get("FINE", "0") == "1":      # every attention / FFN block of the encoder and decoder
    from butd_detr_amd import fused_attention as _fa, attention_blocks as _ab
    counter = [0]
    _orig_block, _orig_ffn = _fa.block, _fa.ffn_block

    def _wrap(fn, tag):
        def g(*a, **k):
            out = fn(*a, **k)
            if torch.cuda.current_stream(dev) != main_stream[0]:
                return out
            counter[0] += 1
            name = f"{tag}{counter[0]}"
            mark("fwd>" + name)
            return first_tensor_map(out, lambda t: _BwdMark.apply(t, name))
        return g
    _fa.block = _wrap(_orig_block, "blk")
    _fa.ffn_block = _wrap(_orig_ffn, "ffn")

if os.environ.get("ENC", "0") == "1":       # both sides of every encoder layer: marks on WHICHEVER stream runs the piece
    def hook_any(obj, attr, name):
        orig = getattr(obj, attr)

        def f(*a, **k):
            q = "main" if torch.cuda.current_stream(dev) == main_stream[0] else "side"
            mark(f"E:{name} start [{q}]")
            out = orig(*a, **k)
            mark(f"E:{name} end [{q}]")
            return out
        setattr(obj, attr, f)
    for i, l in enumerate(m.cross_encoder.layers):
        hook_any(l.self_attention_lang, "forward", f"enc{i} text self-attn")
        hook_any(l.self_attention_visual, "forward", f"enc{i} vis self-attn")
        hook_any(l.cross_layer, "language_branch", f"enc{i} language branch")
        hook_any(l.cross_layer, "vision_branch", f"enc{i} vision branch")

crit = HungarianCriterion()
orig_crit = crit.__call__ if hasattr(crit, "__call__") else None
opt = FlatAdamW(model)
step = GraphedTrainStep(model, opt, criterion=crit)
orig_fb = step._fwd_bwd


def fwd_bwd():
    main_stream[0] = torch.cuda.current_stream(dev)
    names.clear()
    if os.environ.get("FINE", "0") == "1":
        counter[0] = 0
    mark("step start")
    loss = orig_fb()
    mark("step end (gradients packed)")
    return loss


step._fwd_bwd = fwd_bwd
orig_up = step._update


def update():
    mark_names = names          # keep numbering
    r = orig_up()
    mark("update end")
    return r


step._update = update
batches = [synthetic_batch(8, dev, seed=1184 + 50 * i, n_points=50000, tokens=80) for i in range(4)]
replays = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for it in range(4):
    step(batches[it % 4][0], batches[it % 4][1], next_inputs=batches[(it + 1) % 4][0])
torch.cuda.synchronize()
acc = collections.OrderedDict()
tot = []
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
snaps = []
for it in range(replays):
    step(batches[it % 4][0], batches[it % 4][1], next_inputs=batches[(it + 1) % 4][0])
    snaps.append(slots.clone())
ev1.record()
torch.cuda.synchronize()
print(f"free-running replays: {ev0.elapsed_time(ev1) / replays:.3f} ms / step (with {len(names)} marks in the graph)")
n = len(names)
for s in snaps[2:]:
    t = s[:n].cpu().numpy().astype("int64")
    d = (t[1:] - t[:-1]) * 10e-3          # 100 MHz ticks -> us
    tot.append((t[n - 1] - t[0]) * 10e-6)
    for i in range(n - 1):
        acc.setdefault(i, []).append(d[i])
print(f"step start -> update end: {sum(tot) / len(tot):.3f} ms")
if os.environ.get("MARK_NAMES"):
    open(os.environ["MARK_NAMES"], "w").write("\n".join(names) + "\n")
if os.environ.get("ENC", "0") == "1":
    import numpy as np
    rel = np.mean([(sn[:n].cpu().numpy().astype("int64") - int(sn[0])) * 10e-3 for sn in snaps[2:]], axis=0)
    print("encoder pieces, absolute time since step start (us), the stream that ran them:")
    for i in sorted(range(n), key=lambda i: rel[i]):
        if names[i].startswith("E:") or names[i].startswith("fwd>enc") or names[i] == "fwd>pos_embed":
            print(f"  {rel[i]:9.1f}   {names[i]}")
print("  interval (us, mean over replays)      from -> to")
groups = collections.OrderedDict()
for i in range(n - 1):
    v = sum(acc[i]) / len(acc[i])
    print(f"  {v:9.1f}   {names[i]:34s} -> {names[i + 1]}")
    key = names[i + 1].rstrip("0123456789")
    groups.setdefault(key, [0.0, 0])
    groups[key][0] += v; groups[key][1] += 1
print("grouped by the mark that ENDS the interval:")
for k, (v, c) in groups.items():
    print(f"  {v / 1e3:8.3f} ms  x{c:<3d} {k}")
