"""round 4: what do the two prefetch branches (next batch's FPS chain on 8 CUs, next batch's RoBERTa pass) cost the main
stream?  The captured step with the branch bodies replaced by nothing while capturing (their RESULTS stay those of the
warm-up: timing experiment only, the numbers trained on are stale)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, torch, bench
from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, HungarianCriterion, synthetic_batch
dev = torch.device("cuda", 0)
args = argparse.Namespace(backend="auto", queries=256, points=50000, tokens=80, encoder_layers=3)
batches = [synthetic_batch(8, dev, seed=1184 + 50 * i, n_points=50000, tokens=80) for i in range(4)]


def run(skip_fps, skip_text, reps=30, prio=None, variant=None):
    if prio is not None:
        st = torch.cuda.Stream(dev, priority=prio)
        with torch.cuda.stream(st):
            return run(skip_fps, skip_text, reps)
    model, _ = bench.build_model(args, dev)
    step = GraphedTrainStep(model, FlatAdamW(model), criterion=HungarianCriterion())
    o_s, o_t = step._sample_into_next, step._encode_text_into_next
    step._sample_into_next = lambda: None if (skip_fps and torch.cuda.is_current_stream_capturing()) else o_s()
    if variant and variant.startswith("dummy"):     # the language-model branch replaced by N one-workgroup launches
        n_dummy = int(variant[5:])
        cell = torch.zeros(64, device=dev)
        def dummy_text():
            if not torch.cuda.is_current_stream_capturing():
                return o_t()
            for _ in range(n_dummy):
                cell.add_(1.0)
        step._encode_text_into_next = dummy_text
    if variant == "fps1only":      # the branch holds ONLY the first level's sampling (3.4 ms on 8 CUs, 5 launches)
        from butd_detr_amd import pointnet2_utils as pu
        def only_fps1():
            if not torch.cuda.is_current_stream_capturing():
                return o_s()
            keep.append(pu.furthest_point_sample(step._slot.next_pc[..., 0:3].contiguous(), 2048))
        keep = []
        step._sample_into_next = only_fps1
    if variant == "nofps1":        # everything of the branch EXCEPT the first level's sampling (its ~150 small launches)
        from butd_detr_amd import pointnet2_utils as pu
        real, cache = pu.furthest_point_sample, {}
        def fps(xyz, m):
            if m == 2048 and torch.cuda.is_current_stream_capturing() and "i" in cache:
                return cache["i"]
            out = real(xyz, m)
            if m == 2048:
                cache["i"] = out
            return out
        pu.furthest_point_sample = fps
    step._encode_text_into_next = lambda: None if (skip_text and torch.cuda.is_current_stream_capturing()) else o_t()
    for it in range(5):
        step(batches[it % 4][0], batches[it % 4][1], next_inputs=batches[(it + 1) % 4][0])
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import time
    a.record()
    h0 = time.perf_counter()
    per = []
    extra = os.environ.get("EXTRA_EAGER", "0")
    probe = torch.zeros(512, dtype=torch.int64, device=dev)
    for it in range(reps):
        c0 = time.perf_counter()
        step(batches[it % 4][0], batches[it % 4][1], next_inputs=batches[(it + 1) % 4][0])
        if extra == "clone":
            keep_alive = probe.clone()
        elif extra == "add":
            probe.add_(1)
        per.append(time.perf_counter() - c0)
    h1 = time.perf_counter()
    b.record(); torch.cuda.synchronize()
    h2 = time.perf_counter()
    per.sort()
    print(f"host: {reps} calls returned after {(h1 - h0) * 1e3 / reps:.2f} ms per step (median call {per[len(per)//2]*1e3:.2f} ms, min {per[0]*1e3:.2f}), "
          f"drained after {(h2 - h0) * 1e3 / reps:.2f} ms per step")
    return a.elapsed_time(b) / reps


mode = sys.argv[1] if len(sys.argv) > 1 else "base"
if mode == "base":
    print(f"both on, default stream: {run(False, False):.3f} ms / step")
elif mode == "hi":
    print(f"both on, issued on a priority -1 stream: {run(False, False, prio=-1):.3f} ms / step")
elif mode == "lo":
    print(f"both on, issued on a NEW priority 0 stream: {run(False, False, prio=0):.3f} ms / step")
elif mode == "none":
    print(f"both off: {run(True, True):.3f} ms / step")
elif mode == "fps1only":
    print(f"branch = level-1 sampling only, RoBERTa off: {run(False, True, reps=60, variant='fps1only'):.3f} ms / step")
elif mode == "nofps1":
    print(f"branch = plan without level-1 sampling, RoBERTa off: {run(False, True, reps=60, variant='nofps1'):.3f} ms / step")
elif mode.startswith("dummy"):
    print(f"sampling off, language model = {mode[5:]} one-workgroup launches: {run(True, False, reps=60, variant=mode):.3f} ms / step")
elif mode == "nofps":
    print(f"FPS chain off, RoBERTa on: {run(True, False, reps=60):.3f} ms / step")
elif mode == "notext":
    print(f"FPS chain on, RoBERTa off: {run(False, True, reps=60):.3f} ms / step")
elif mode == "base60":
    print(f"both on: {run(False, False, reps=60):.3f} ms / step")
elif mode == "none60":
    print(f"both off: {run(True, True, reps=60):.3f} ms / step")
if mode.startswith("two"):
    # two GraphedTrainStep objects (two sets of graph executables over the same model / optimizer), alternating: does the
    # host enqueue of a step wait for the previous launch of the SAME executable?
    import time
    skip = mode == "two_none"
    model, _ = bench.build_model(args, dev)
    opt = FlatAdamW(model)
    steps = [GraphedTrainStep(model, opt, criterion=HungarianCriterion()) for _ in range(2)]
    for st in steps:
        o_s, o_t = st._sample_into_next, st._encode_text_into_next
        st._sample_into_next = (lambda o=o_s: None if (skip and torch.cuda.is_current_stream_capturing()) else o())
        st._encode_text_into_next = (lambda o=o_t: None if (skip and torch.cuda.is_current_stream_capturing()) else o())
    def call(it):
        steps[it % 2](batches[it % 4][0], batches[it % 4][1], next_inputs=batches[(it + 2) % 4][0])
    for it in range(8):
        call(it)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); h0 = time.perf_counter()
    for it in range(60):
        call(it)
    h1 = time.perf_counter(); b.record(); torch.cuda.synchronize()
    print(f"{mode}: host returned after {(h1 - h0) * 1e3 / 60:.2f} ms per step; {a.elapsed_time(b) / 60:.3f} ms / step")
