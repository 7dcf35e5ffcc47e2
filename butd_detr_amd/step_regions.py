"""In-situ timeline of the captured training step WITHOUT a profiler: where the step's time goes, region by region,
and what each region achieves against the fp32 matrix peak while it runs next to everything else in the step.

Region boundaries are one-wave kernels that write the device wall clock (``butd_timeline_mark``, s_memrealtime,
include/butd_graph.h) when the MAIN stream reaches them: a forward mark after a module's forward, a backward mark when
autograd is about to run that module's backward (an identity Function on its output).  The marks are captured into the
step's hipGraph with everything else; after N free-running replays the intervals are averaged.  A mark is one launch on
the main queue (~3.5 us): ~50 marks inflate the step by ~0.2 ms, which ``measure()`` reports next to the intervals.

The matrix work of an interval is counted while the step is captured: every grouped-GEMM call (2 M N K per problem) and
every attention-core launch (4 B H Lq Lk D forward, 2.5 x backward) adds to the interval that is open at that moment.
The set-abstraction products that do not go through the grouped GEMM (csrc/sa_last_bwd.hip) are added analytically by
the caller (bench.py).

Regions follow the reference's call stack (models/bdetr.py:193-319):
    backbone   Pointnet2Backbone: sa1..sa4, fp1, fp2                                    (backbone_module.py:111-160)
    encoder    pos_embed, text projector, BiEncoder layers                               (bdetr.py:212-236)
    decoder    points_obj_cls, gsample, decoder_query_proj, proposal head, the six BiDecoderLayers, their prediction
               heads and the contrastive projection of the queries                       (bdetr.py:238-317)
    criterion  from the end of the last head's forward to the moment autograd reaches the last decoder layer:
               compute_hungarian_loss forward + backward and the last head's backward    (models/losses.py:519-617)
"""
import collections

import torch

from . import _hiplib

MAX_MARKS = 512
FP32_MATRIX_PEAK_TF = 157.3

_REGION_OF = (("sa", "backbone"), ("fp", "backbone"),
              ("pos_embed", "encoder"), ("text_proj", "encoder"), ("enc", "encoder"), ("proj_txt", "encoder"),
              ("objcls", "decoder"), ("gsample", "decoder"), ("query_proj", "decoder"), ("proposal_head", "decoder"),
              ("dec", "decoder"), ("head", "decoder"), ("proj_img", "decoder"))


def region_of(module_name):
    for prefix, region in _REGION_OF:
        if module_name.startswith(prefix):
            return region
    return "other"


class _BwdMark(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, marks, name):
        ctx.marks, ctx.name = marks, name
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.marks.mark("bwd<" + ctx.name)        # autograd is about to run this module's backward
        return g, None, None


def _mark_first(out, fn):
    if torch.is_tensor(out):
        return fn(out) if out.requires_grad else out
    if isinstance(out, tuple):
        done, res = False, []
        for o in out:
            if not done and torch.is_tensor(o) and o.requires_grad:
                res.append(fn(o))
                done = True
            else:
                res.append(o)
        return tuple(res)
    return out


class StepMarks:
    """Marks on a ``BeaUTyDETR`` + ``GraphedTrainStep`` pair.  ``install()`` hooks the modules and the step; every
    (re)capture of the step then carries the marks; ``measure()`` replays and reads them."""

    def __init__(self, model, step, device):
        self.model, self.step, self.device = model, step, torch.device(device)
        self.lib = _hiplib.load()
        self.slots = torch.zeros(MAX_MARKS, dtype=torch.int64, device=self.device)
        self.names, self.flops = [], collections.defaultdict(float)
        self.main = None
        self._undo = []

    # -- marks
    def mark(self, name):
        if len(self.names) >= MAX_MARKS:
            return
        self.names.append(name)
        self.lib.butd_timeline_mark(self.slots.data_ptr(), len(self.names) - 1,
                                    torch.cuda.current_stream(self.device).cuda_stream)

    def _hook(self, mod, name):
        orig = mod.forward

        def forward(*a, **k):
            out = orig(*a, **k)
            if self.main is None or torch.cuda.current_stream(self.device) != self.main:
                return out
            self.mark("fwd>" + name)
            return _mark_first(out, lambda t: _BwdMark.apply(t, self, name))
        mod.forward = forward
        self._undo.append(lambda: mod.__dict__.pop("forward", None))

    def _count(self, flops):
        if self.main is not None and torch.cuda.current_stream(self.device) == self.main:
            self.flops[len(self.names) - 1] += flops       # the interval that the last mark opened

    def install(self):
        from . import fused_attention as fa, fused_mlp as fmlp, fused_sa as fsa
        m = self.model
        for n in ("sa1", "sa2", "sa3", "sa4", "fp1", "fp2"):
            self._hook(getattr(m.backbone_net, n), n)
        for i, l in enumerate(m.cross_encoder.layers):
            self._hook(l, f"enc{i}")
        for i, l in enumerate(m.decoder):
            self._hook(l, f"dec{i}")
        for i, l in enumerate(m.prediction_heads):
            self._hook(l, f"head{i}")
        for attr, name in (("proposal_head", "proposal_head"), ("points_obj_cls", "objcls"), ("pos_embed", "pos_embed"),
                           ("text_projector", "text_proj"), ("gsample_module", "gsample"),
                           ("decoder_query_proj", "query_proj"), ("contrastive_align_projection_image", "proj_img"),
                           ("contrastive_align_projection_text", "proj_txt")):
            if getattr(m, attr, None) is not None:
                self._hook(getattr(m, attr), name)
        orig_gemm = fa._gemm

        def counted(problems, ref):
            self._count(sum(2.0 * p.M * p.N * p.K for p in problems if p.fold_src is None))
            orig_gemm(problems, ref)
        fa._gemm = fsa._gemm = fmlp._gemm = counted
        prev_hook = fa._work_hook[0]
        fa._work_hook[0] = lambda kind, flops: self._count(flops)

        def undo_counts():
            fa._gemm = fsa._gemm = fmlp._gemm = orig_gemm
            fa._work_hook[0] = prev_hook
        self._undo.append(undo_counts)
        step = self.step
        orig_fb, orig_up = step._fwd_bwd, step._update

        def fwd_bwd():
            self.main = torch.cuda.current_stream(self.device)
            del self.names[:]
            self.flops.clear()
            self.mark("step start")
            loss = orig_fb()
            self.mark("step end (gradients packed)")
            return loss

        def update():
            r = orig_up()
            self.mark("update end")
            return r
        step._fwd_bwd, step._update = fwd_bwd, update
        self._undo.append(lambda: (step.__dict__.pop("_fwd_bwd", None), step.__dict__.pop("_update", None)))
        return self

    def remove(self):
        for fn in reversed(self._undo):
            fn()
        del self._undo[:]

    # -- measurement
    def measure(self, batches, replays=12, skip=2):
        """Free-running replays over ``batches`` (the step must already be captured with the marks: call it a few times
        first) -> {"ms_per_step", "intervals": [(from, to, us, flops)], "regions": {...}}."""
        step, n = self.step, len(batches)
        torch.cuda.synchronize(self.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        snaps = []
        e0.record()
        for it in range(replays):
            step(batches[it % n][0], batches[it % n][1], next_inputs=batches[(it + 1) % n][0])
            snaps.append(self.slots.clone())
        e1.record()
        torch.cuda.synchronize(self.device)
        k = len(self.names)
        ticks = torch.stack(snaps[skip:])[:, :k].cpu().double()          # 100 MHz ticks
        us = ((ticks[:, 1:] - ticks[:, :-1]) * 1e-2).mean(0).tolist()
        intervals = [(self.names[i], self.names[i + 1], us[i], self.flops.get(i, 0.0)) for i in range(k - 1)]
        return {"ms_per_step": e0.elapsed_time(e1) / replays, "marks": k, "intervals": intervals,
                "regions": summarize(intervals)}


def summarize(intervals):
    """Intervals -> regions.  A forward interval belongs to the module whose mark ENDS it (the mark fires after the
    module's forward), a backward interval to the module whose mark STARTS it (the mark fires when its backward
    begins); the interval between the last forward mark and the first backward mark is the criterion."""
    acc = collections.OrderedDict()

    def add(region, us, flops):
        r = acc.setdefault(region, [0.0, 0.0])
        r[0] += us
        r[1] += flops

    for a, b, us, flops in intervals:
        if b.startswith("fwd>"):
            add(region_of(b[4:]) + " fwd", us, flops)
        elif a.startswith("bwd<"):
            add(region_of(a[4:]) + " bwd", us, flops)
        elif a.startswith("fwd>") and b.startswith("bwd<"):
            add("criterion fwd+bwd", us, flops)
        elif b == "update end":
            add("clip + AdamW", us, flops)
        else:                                     # (the last backward interval ends at "step end")
            add(region_of(a[4:]) + " bwd" if a.startswith("bwd<") else "other", us, flops)
    out = collections.OrderedDict()
    for name, (us, flops) in acc.items():
        out[name] = {"ms": round(us * 1e-3, 3), "gflop": round(flops * 1e-9, 2)}
    for region in ("backbone", "encoder", "decoder"):
        us = sum(v[0] for k, v in acc.items() if k.startswith(region))
        flops = sum(v[1] for k, v in acc.items() if k.startswith(region))
        if us > 0:
            out[region] = {"ms": round(us * 1e-3, 3), "gflop": round(flops * 1e-9, 2),
                           "tflops": round(flops / (us * 1e-6) / 1e12, 2),
                           "frac_of_fp32_matrix_peak": round(flops / (us * 1e-6) / 1e12 / FP32_MATRIX_PEAK_TF, 4)}
    return out


def format_intervals(result):
    lines = [f"free-running replays: {result['ms_per_step']:.3f} ms / step (with {result['marks']} marks in the graph)",
             "  interval (us, mean over replays)   GFLOP   from -> to"]
    for a, b, us, flops in result["intervals"]:
        lines.append(f"  {us:9.1f}  {flops * 1e-9:8.2f}   {a:34s} -> {b}")
    lines.append("regions:")
    for k, v in result["regions"].items():
        extra = (f"  {v['tflops']:6.1f} TF = {v['frac_of_fp32_matrix_peak']:.3f} of the fp32 matrix peak"
                 if "tflops" in v else "")
        lines.append(f"  {v['ms']:8.3f} ms  {v['gflop']:8.2f} GF  {k}{extra}")
    return "\n".join(lines)
