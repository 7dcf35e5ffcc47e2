#!/usr/bin/env python
"""bench.py -- scenes/sec fwd+bwd of the BUTD-DETR hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = forward + surrogate loss + backward + grad-clip + AdamW update of ``BeaUTyDETR`` on one
synthetic batch per GPU (``--batch`` scenes of 50 000 points, 256 queries, 80 tokens, 132 box slots;
BASELINE configs[2]/[3]: 8 scenes per GPU, weak scaling), inputs resident in HBM before the timed
region.  Rank 0 prints ONE JSON line with the throughput, the roofline of the dominant hand-written
kernel (HIP-event timed on its own stream) and, at N=1, a CPU baseline (oracle ops + torch CPU fp32
on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
FP32_MATRIX_PEAK_TF = 157.3
BF16_MATRIX_PEAK_TF = 2500.0   # dense, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="scenes per GPU")
    ap.add_argument("--points", type=int, default=50000)
    ap.add_argument("--queries", type=int, default=256)
    ap.add_argument("--tokens", type=int, default=80)
    ap.add_argument("--backend", default=os.environ.get("BUTD_ATTENTION_BACKEND", "auto"),
                    choices=["auto", "torch", "hip"])
    ap.add_argument("--criterion", default="hungarian", choices=["hungarian", "surrogate"],
                    help="hungarian: the reference's criterion (models/losses.py) with the assignment on the "
                         "device; surrogate: dense stand-in without matching")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="plain eager launches + DDP instead of hipGraph replay")
    ap.add_argument("--cpu-scenes", type=int, default=8,
                    help="scenes per step of the CPU baseline (8 = the batch of the GPU line, BASELINE.md section 3)")
    ap.add_argument("--encoder-layers", type=int, default=3,
                    help="BiEncoder depth: 3 = the reference (models/bdetr.py:104); 6 = the extra row BASELINE "
                         "configs[2] words as '6-layer BiEncoder'")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="arithmetic of the grouped products (projections, FFN, 1x1-conv chains, their gradients): "
                         "f32 = the reference's precision (the headline); bf16 = operands rounded to bf16 on the "
                         "matrix cores, fp32 accumulation (BASELINE configs[3]).  The default run reports the bf16 "
                         "mode as an extra field next to the fp32 headline")
    ap.add_argument("--no-bf16-row", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra operating points and kernel microbenchmarks (bf16 row, utterance cache, "
                         "6-layer encoder, attention / ball-query / matcher sections): the command the committed "
                         "rocprofv3 kernel statistics under profiles/ are taken with")
    ap.add_argument("--roofline-child", action="store_true",
                    help="internal: run only the captured training step (warm-up + steps) and print the grouped GEMM's "
                         "algorithmic work per step; the parent runs this under rocprofv3 --kernel-trace --stats")
    ap.add_argument("--overlap-exchange", dest="overlap_exchange", action="store_true", default=None,
                    help="N > 1: the two-piece capture whose decoder-side gradient bucket is all-reduced under the encoder / "
                         "backbone backward, instead of the default one graph + ONE all-reduce of the whole packed buffer "
                         "(opt-in until a run on >= 2 GPUs has compared the loss trajectories; BUTD_OVERLAP_EXCHANGE=1 "
                         "does the same)")
    ap.add_argument("--distinct-batches", type=int, default=4,
                    help="the timed steps rotate over this many different synthetic batches (scene geometry decides "
                         "how much the pruned FPS and the ball query do; one batch replayed is its best case)")
    ap.add_argument("--max-targets", type=int, default=16,
                    help="target boxes per scene of the synthetic ground truth: 1..16 (grounding splits) or, "
                         "e.g., 132 = 66..132 per scene (the detection split fills the 132 slots)")
    return ap.parse_args()


def build_model(args, device):
    import warnings
    from butd_detr_amd import attention_blocks
    from butd_detr_amd.bdetr import BeaUTyDETR
    from butd_detr_amd.offline_text import offline_factory
    # "auto" IS the hand-written gfx950 path: a missing / incomplete HIP library raises here (no silent
    # fallback); the stock-torch blocks only run when asked for by name (--backend torch, the A/B leg)
    backend = "hip" if args.backend == "auto" else args.backend
    attention_blocks.set_backend(backend)
    attention_blocks.set_strict(backend == "hip")     # a level that would fall back to stock torch raises
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = BeaUTyDETR(num_class=256, num_obj_class=485, input_feature_dim=3,
                           num_queries=args.queries, num_decoder_layers=6,
                           self_position_embedding="loc_learned", contrastive_align_loss=True,
                           butd=True, self_attend=True, text_encoder_factory=offline_factory(0),
                           num_encoder_layers=args.encoder_layers)
    return model.to(device).train(), backend


def gemm_work(step_fn):
    """Algorithmic work of the grouped GEMM launches of ONE eager training step (no timing): number of launches,
    sum over their problems of 2*M*N*K flops and of (M*K + N*K + M*N)*4 bytes (every operand and the result once)."""
    from butd_detr_amd import fused_attention as fa
    import butd_detr_amd.fused_mlp as fmlp
    import butd_detr_amd.fused_sa as fsa
    work = {"launches": 0, "flops": 0.0, "bytes": 0.0, "attention_flops": 0.0, "attention_launches": 0}
    orig = fa._gemm

    def counted(problems, ref):
        work["launches"] += 1
        work["flops"] += sum(2.0 * p.M * p.N * p.K for p in problems)
        work["bytes"] += sum(4.0 * (p.M * p.K + p.N * p.K + p.M * p.N) for p in problems)
        orig(problems, ref)

    def attention(kind, flops):         # every attention-core call of the step (fused_attention._count_attention)
        work["attention_flops"] += flops
        work["attention_launches"] += 1

    fa._gemm = fsa._gemm = fmlp._gemm = counted
    prev_hook, fa._work_hook[0] = fa._work_hook[0], attention
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        fa._gemm = fsa._gemm = fmlp._gemm = orig
        fa._work_hook[0] = prev_hook
    return work


def _event_timed_gemm_work(step_fn):
    """gemm_work() with a HIP event pair around every launch -> (work, average ms per launch)."""
    from butd_detr_amd import fused_attention as fa
    import butd_detr_amd.fused_mlp as fmlp
    import butd_detr_amd.fused_sa as fsa
    stream = torch.cuda.current_stream()
    work, pairs, orig = {"launches": 0, "flops": 0.0, "bytes": 0.0}, [], fa._gemm

    def timed(problems, ref):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        work["launches"] += 1
        work["flops"] += sum(2.0 * p.M * p.N * p.K for p in problems)
        work["bytes"] += sum(4.0 * (p.M * p.K + p.N * p.K + p.M * p.N) for p in problems)
        e0.record(stream)
        orig(problems, ref)
        e1.record(stream)
        pairs.append((e0, e1))

    fa._gemm = fsa._gemm = fmlp._gemm = timed
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        fa._gemm = fsa._gemm = fmlp._gemm = orig
    return work, sum(a.elapsed_time(b) for a, b in pairs) / max(len(pairs), 1)


def _rocprof_kernel_stats(argv, kernel, timeout=600):
    """Run ``python bench.py <argv>`` under ``rocprofv3 --kernel-trace --stats`` and return (calls, total ns) of the
    kernels whose name contains ``kernel`` plus the child's last stdout line.  Pure kernel durations from the
    dispatch timestamps -- the same source as the committed profiles/*_kernel_stats.csv."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    out_dir = tempfile.mkdtemp(prefix="butd_rocprof_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    # the child is a plain one-GPU run on THIS rank's device, whatever launched the parent (torchrun at N > 1)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR",
              "MASTER_PORT", "TORCHELASTIC_RUN_ID", "BUTD_BENCH_FORCE_DIST", "BUTD_FORCE_COLLECTIVE"):
        env.pop(k, None)
    dev = torch.cuda.current_device()
    visible = [d for d in env.get("HIP_VISIBLE_DEVICES", "").split(",") if d]
    env["HIP_VISIBLE_DEVICES"] = visible[dev] if dev < len(visible) else str(dev)
    env.pop("BUTD_BENCH_ONE_GPU", None)
    cmd = [exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", out_dir, "-o", "rf", "--",
           sys.executable, os.path.join(ROOT, "bench.py")] + argv
    try:
        res = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
        files = glob.glob(os.path.join(out_dir, "**", "*kernel_stats.csv"), recursive=True)
        if res.returncode != 0 or not files:
            raise RuntimeError(f"rocprofv3 child failed (rc {res.returncode}): {res.stderr[-400:]}")
        calls, total = 0, 0
        _LAST_CHILD_ROWS.clear()
        for row in csv.DictReader(open(files[0])):
            _LAST_CHILD_ROWS.append((row["Name"], int(row["Calls"]), int(row["TotalDurationNs"])))
            if kernel in row["Name"]:
                calls += int(row["Calls"])
                total += int(row["TotalDurationNs"])
        lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
        return calls, total, (json.loads(lines[-1]) if lines else {})
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


_LAST_CHILD_ROWS = []        # (kernel name, calls, total ns) of the last rocprofv3 child: gemm_roofline's trace, reused below


def sa_linear_roofline(args, gemm):
    """The set-abstraction products that left ``gemm_kernel`` in round 4 (csrc/sa_last_bwd.hip: the last layer + max-pool
    forward without Z3, its backward by linearity, SA1's layer 2 + layer 1 backward in one pass) -- durations from the SAME
    child trace as ``roofline``; flops = what these kernels execute on the matrix cores (forward 2 P C2 C3; backward
    4 P C2^2: H.A and the Gram matrix; SA1's lower pass 4 P C^2), bytes = their algorithmic HBM traffic (Z2 read per pass,
    the gated gradient written once, SA1's lower pass reads g2, Z2, Z1, X).  Also the matrix-core rate of gemm_kernel and
    these kernels TOGETHER: the figure comparable with round 3's ``roofline.frac`` (the 10^5..10^6-row products, the most
    efficient launches of gemm_kernel, are the ones that moved)."""
    if not _LAST_CHILD_ROWS or not gemm.get("launches_per_step"):
        return None
    names = ("sa_last_fwd_kernel", "sa_last_fused_kernel", "sa_last_mfma_kernel", "sa_last_sparse_kernel", "sa_mid_first_kernel",
             "sa_l12_fwd_kernel", "sa_mid_wide_kernel")
    gemm_calls = sum(c for n, c, _ in _LAST_CHILD_ROWS if "gemm_kernel" in n)
    steps = (sum(c for n, c, _ in _LAST_CHILD_ROWS if "lsap_kernel" in n)
             or gemm_calls / gemm["launches_per_step"])       # steps in the child's trace
    per = {k: sum(t for n, _, t in _LAST_CHILD_ROWS if k in n) / steps * 1e-6 for k in names}     # ms per step
    ms = sum(per.values())
    if ms <= 0:
        return None
    B, npts = args.batch, args.points
    levels = [(B * 2048 * 64, 64, 128), (B * 1024 * 32, 128, 256), (B * 512 * 16, 128, 256), (B * 256 * 16, 128, 256)]
    P1 = levels[0][0]
    flops = sum(2.0 * P * c2 * c3 + 4.0 * P * c2 * c2 for P, c2, c3 in levels) + 4.0 * P1 * 64 * 64
    bytes_ = sum(4.0 * P * c2 * (1 + 2) for P, c2, _ in levels)           # forward reads Z2; backward reads Z2, writes g2
    bytes_ += sum(4.0 * P * c2 * 2 for P, c2, _ in levels[1:])            # 128-wide levels: two-kernel backward (O round trip)
    if per["sa_l12_fwd_kernel"] > 0:      # SA1's first two layers forward without Z1; the lower pass recomputes z1 twice
        flops += 2.0 * P1 * 64 * 64 + 3 * 2.0 * P1 * 64 * 8
        bytes_ += 4.0 * P1 * (8 + 64) + 4.0 * P1 * (2 * 64 + 8)
    else:
        bytes_ += 4.0 * P1 * (3 * 64 + 8)                                 # SA1 lower pass reading Z1
    if per["sa_mid_wide_kernel"] > 0:     # round 5: SA2-4's layer 2 backward in one pass (dW2 and dH1: 4 P C^2; g2, Z2, Z1 read, g1 written)
        flops += sum(4.0 * P * c2 * c2 for P, c2, _ in levels[1:])
        bytes_ += sum(4.0 * P * c2 * 4 for P, c2, _ in levels[1:])
    both_ms = ms + gemm["ms_per_step_in_kernel"]
    both = (flops + gemm["algorithmic_flops_per_step"]) / (both_ms * 1e-3) / 1e12
    return {"kernel": "sa_last_fwd / sa_last_fused / sa_last_mfma + sa_last_sparse / sa_mid_first / sa_l12_fwd / sa_mid_wide (set-abstraction "
                      "last layer + max-pool by linearity, SA1 lower layers, SA2-4 middle layer backward: the products that left gemm_kernel)",
            "bound": "mfma", "achieved": round(flops / (ms * 1e-3) / 1e12, 2), "peak": FP32_MATRIX_PEAK_TF, "unit": "TFLOP/s",
            "frac": round(flops / (ms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TF, 4), "traffic": None,
            "ms_per_step_in_kernels": round(ms, 3), "ms_per_kernel": {k: round(v, 3) for k, v in per.items()},
            "matrix_flops_per_step": flops, "algorithmic_bytes_per_step": bytes_,
            "hbm_gbps_algorithmic": round(bytes_ / (ms * 1e-3) / 1e9, 1),
            "with_gemm_kernel": {"ms_per_step": round(both_ms, 3), "achieved": round(both, 2),
                                 "frac": round(both / FP32_MATRIX_PEAK_TF, 4),
                                 "note": "all fp32 matrix-core product kernels of the step (attention core excluded): "
                                         "comparable with round 3's roofline.frac"}}


def gemm_roofline(args, fallback_step=None):
    """Dominant hand-written kernel of the step: ``gemm_kernel`` (fp32 MFMA grouped GEMM: every projection / FFN /
    1x1-conv product of the attention stack and of the set-abstraction MLPs, all their gradient products, the Conv1d
    chains of the heads; ~45 % of the GPU time of a step).  Duration: the captured training step is run once more in
    a CHILD process under ``rocprofv3 --kernel-trace --stats`` (this command with --roofline-child) and the kernel's
    dispatch durations are summed -- in situ, next to the concurrent sampling / language-model branches, no event
    overhead to calibrate away; the same method as profiles/r03_hip_bench_kernel_stats.csv, which must agree.
    achieved = algorithmic flops per step (2*M*N*K over the problems of every launch, counted in the child)
    / (average launch duration x launches per step)."""
    child = ["--roofline-child", "--steps", "12", "--warmup", "3", "--batch", str(args.batch), "--points",
             str(args.points), "--queries", str(args.queries), "--tokens", str(args.tokens), "--encoder-layers",
             str(args.encoder_layers), "--max-targets", str(args.max_targets), "--criterion", args.criterion,
             "--dtype", args.dtype, "--distinct-batches", str(args.distinct_batches)]
    source = None
    try:
        if os.environ.get("BUTD_BENCH_NO_CHILD") == "1":   # the parent itself runs under a profiler (scratch/*.sh)
            raise RuntimeError("BUTD_BENCH_NO_CHILD=1")
        calls, total_ns, work = _rocprof_kernel_stats(child, "gemm_kernel")
        if not calls or not work.get("launches"):
            raise RuntimeError("no gemm_kernel dispatches in the child's trace")
        avg_ms = total_ns / calls * 1e-6
    except Exception as exc:  # noqa: BLE001 -- no profiler on this box: HIP events around every launch (an upper bound:
        # each pair also times the gap between the event packets and the kernel)
        if fallback_step is None:
            raise
        work, avg_ms = _event_timed_gemm_work(fallback_step)
        calls, total_ns = work["launches"], 0
        source = f"HIP event pairs around every launch of one eager step (rocprofv3 child unavailable: {exc!r:.120})"
    # steps in the child's trace = launches of the assignment kernel (one per step: criterion = hungarian); the grouped
    # calls counted in Python are NOT the kernel launches (a call with float4-aligned and element-wise staged problems is
    # two launches, the riders of the deterministic split-K folds none, their final flush one more): round 4 multiplied
    # the average launch by the calls
    steps = sum(c for n, c, _ in _LAST_CHILD_ROWS if "lsap_kernel" in n) if source is None else 0
    if steps > 0:
        launches = calls / steps
        ms_step = total_ns / steps * 1e-6
    else:
        launches = work["launches"]
        ms_step = avg_ms * work["launches"]
    achieved = work["flops"] / (ms_step * 1e-3) / 1e12
    traffic, traffic_source = _pmc_traffic("gemm_kernel", with_source=True)
    return {"kernel": "gemm_kernel (grouped fp32 MFMA GEMM, all %.1f launches of one training step)" % launches,
            "bound": "mfma", "achieved": round(achieved, 2), "peak": FP32_MATRIX_PEAK_TF, "unit": "TFLOP/s",
            "frac": round(achieved / FP32_MATRIX_PEAK_TF, 4), "traffic": traffic,
            "traffic_source": traffic_source,
            "launches_per_step": round(launches, 2), "grouped_calls_per_step": work["launches"],
            "avg_launch_ms": round(avg_ms, 5),
            "ms_per_step_in_kernel": round(ms_step, 3),
            "duration_source": source or ("rocprofv3 --kernel-trace --stats of the captured step in a child process "
                                          "(%d launches over %d steps profiled)" % (calls, steps)),
            "algorithmic_flops_per_step": work["flops"], "algorithmic_bytes_per_step": work["bytes"],
            "attention_flops_per_step": work.get("attention_flops"), "attention_calls_per_step": work.get("attention_launches"),
            "steps_in_trace": steps,
            "algorithmic_bytes_per_launch": round(work["bytes"] / max(launches, 1)),
            "traffic_over_algorithmic": (round(traffic / (work["bytes"] / max(launches, 1)), 3)
                                         if traffic else None)}


def attention_roofline(batch, reps=10, bf16=False):
    """The attention core on the encoder's visual self-attention shape (B x 8 heads, 1024 x 1024, head
    dim 36, dropout 0.1): forward and backward launches timed with HIP events on the launch stream;
    algorithmic FLOPs = 4*Lq*Lk*D per head forward, 2.5x that backward."""
    from butd_detr_amd import _hiplib, fused_attention as fa
    lib = _hiplib.load()
    B, H, D, L = batch, 8, 36, 1024
    E = H * D
    dev = torch.device("cuda", torch.cuda.current_device())
    q, k, v, do = (torch.randn(B, L, E, device=dev) for _ in range(4))
    out, dq, dk, dv = (torch.empty(B, L, E, device=dev) for _ in range(4))
    lse, delta = torch.empty(B, H, L, device=dev), torch.empty(B, H, L, device=dev)
    ctr = fa.rng_counter(dev).data_ptr()
    stream = torch.cuda.current_stream()

    fwd_fn = lib.butd_attention_fwd_bf16 if bf16 else lib.butd_attention_fwd
    bwd_fn = lib.butd_attention_bwd_bf16 if bf16 else lib.butd_attention_bwd

    def fwd():
        return fwd_fn(B, H, L, L, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), None, out.data_ptr(),
                                      lse.data_ptr(), 0.1, 7, ctr, stream.cuda_stream)

    # the backward the product runs for this shape (fused_attention._attention_backward): the one-pass kernel + its
    # slab fold when the library serves the shape (fp32), else the dQ walk + the dK/dV walk
    need = int((lib.butd_attention_bwd_long_keys_bf16_scratch if bf16 else lib.butd_attention_bwd_long_keys_scratch)(B, H, L, L, D, 0))
    ws = torch.empty(max(need, 1), device=dev)
    one_pass = lib.butd_attention_bwd_long_keys_bf16 if bf16 else lib.butd_attention_bwd_long_keys

    def bwd():
        if need >= 0:
            return one_pass(B, H, L, L, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), None,
                                                    out.data_ptr(), do.data_ptr(), lse.data_ptr(), dq.data_ptr(),
                                                    dk.data_ptr(), dv.data_ptr(), 0, 0, 1.0, 0.1, 7, ctr, ws.data_ptr(),
                                                    need, stream.cuda_stream)
        return bwd_fn(B, H, L, L, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), None, out.data_ptr(),
                                      do.data_ptr(), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), dk.data_ptr(),
                                      dv.data_ptr(), 0, 0, 1.0, 0.1, 7, ctr, stream.cuda_stream)

    def timed(fn):
        assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    ms_f, ms_b = timed(fwd), timed(bwd)
    flops_f = 4.0 * L * L * D * H * B
    achieved = (flops_f * 3.5) / ((ms_f + ms_b) * 1e-3) / 1e12
    peak = BF16_MATRIX_PEAK_TF if bf16 else FP32_MATRIX_PEAK_TF
    return {"kernel": "attn_fwd / %s%s (B=%d, 8 heads, 1024x1024, head dim 36)"
                      % ("attn_bwd_longk + attn_dq_fold (one pass)" if need >= 0 else "attn_bwd_dq / attn_bwd_dkv",
                         " <bf16 matrix steps>" if bf16 else "", B),
            "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": _pmc_traffic("attn_fwd_kernel"),
            "traffic_bwd": _pmc_traffic("attn_bwd_longk_kernel") if need >= 0 else _pmc_traffic("attn_bwd_dkv_kernel"),
            "fwd_ms": round(ms_f, 4), "bwd_ms": round(ms_b, 4)}


def _pmc_traffic(kernel, with_source=False):
    """HBM bytes per launch from the newest committed PMC profile (profiles/r06_pmc.json, else r05 / r04 / r03 / r02), or None.
    NOT measured in this run: the counters need their own rocprofv3 --pmc passes (scratch/pmc.sh); ``traffic_source``
    in the record names the file."""
    for name in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json"):
        try:
            v = json.load(open(os.path.join(ROOT, "profiles", name))).get(kernel, {}).get("hbm_bytes_per_launch")
        except Exception:
            v = None
        if v is not None:
            src = f"profiles/{name} (separate rocprofv3 --pmc passes of the same command; not collected in this run)"
            return (v, src) if with_source else v
    return (None, None) if with_source else None


def ball_query_roofline(inputs, steps=20):
    """SA1 ball query (M=2048 centres x N points, ns=64): algorithmic bytes = M*N*12 + M*ns*4 + M*12
    per scene; duration from HIP events on the launch stream.  Measured twice:
    the product path (pointnet2_ext.ball_query -> butd_ball_query_ws: uniform grid + per-centre hit bitmap,
    which only visits the 27 cells around a centre) and the streaming kernel (butd_ball_query) that
    evaluates every (centre, point) pair."""
    from butd_detr_amd import _hiplib, pointnet2_ext as ext
    lib = _hiplib.load()
    xyz = inputs["point_clouds"][..., :3].contiguous()
    b, n = xyz.shape[0], xyz.shape[1]
    inds = ext.furthest_point_sampling(xyz, 2048)
    new_xyz = torch.gather(xyz, 1, inds.long()[..., None].expand(-1, -1, 3)).contiguous()
    stream = torch.cuda.current_stream()
    idx = torch.empty((b, 2048, 64), dtype=torch.int32, device=xyz.device)

    def timed(fn):
        for _ in range(3):
            fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
               for _ in range(steps)]
        for s, e in evs:
            s.record(stream)
            fn()
            e.record(stream)
        torch.cuda.synchronize()
        return sum(s.elapsed_time(e) for s, e in evs) / steps

    ms = timed(lambda: ext.ball_query(new_xyz, xyz, 0.2, 64))
    ms_stream = timed(lambda: lib.butd_ball_query(b, n, 2048, 0.2, 64, new_xyz.data_ptr(), xyz.data_ptr(),
                                                  idx.data_ptr(), stream.cuda_stream))
    alg_bytes = b * (2048 * n * 12 + 2048 * 64 * 4 + 2048 * 12)
    pair_tests = b * 2048.0 * n
    fps_evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for s, e in fps_evs:
        s.record(stream)
        ext.furthest_point_sampling(xyz, 2048)
        e.record(stream)
    torch.cuda.synchronize()
    fps_ms = sum(s.elapsed_time(e) for s, e in fps_evs) / len(fps_evs)
    pruned = int(lib.butd_ball_query_workspace_bytes(b, n, 2048)) > 0
    # what the pruned path has to move at least: the cloud once (grid build: read xyz, write the sorted
    # (x, y, z, index) records), the records of the 27 cells around every centre (~170 candidates), the index rows
    moved = b * (n * 12 + n * 16 + 2048 * 170 * 16 + 2048 * 64 * 4 + 2048 * 12)
    traffic = _pmc_traffic("bq_grid_query" if pruned else "ball_query_kernel")
    return {"kernel": ("butd_ball_query_ws: bq_grid_bbox/count/scan/scatter/query" if pruned else "ball_query_kernel")
                      + " (SA1: 2048 centres x %d points, nsample 64, B=%d)" % (n, b),
            "bound": "latency (device-scope atomics of the grid build + dependent loads of the query), not hbm",
            "avg_launch_ms": round(ms, 5), "pair_tests_per_s": round(pair_tests / (ms * 1e-3), 1),
            "logical_pair_tests_per_launch": pair_tests,
            "compulsory_bytes_per_launch": moved,
            "achieved": round(moved / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
            "note": "north_star asks for >= 50 %% of HBM peak on ball query; that bound belongs to the streaming "
                    "algorithm (every centre x every point, SURVEY section 8(d): M*N*12 B = %d bytes per launch).  "
                    "The exact grid-pruned path visits only the 27 cells around a centre: it moves ~%.0f MB in "
                    "%.0f us and is bound by atomics / dependent-load latency, 5.7x faster than streaming"
                    % (alg_bytes, moved / 1e6, ms * 1e3),
            "streaming_kernel": {"kernel": "ball_query_kernel<8>", "avg_launch_ms": round(ms_stream, 5),
                                 "pair_tests_per_s": round(pair_tests / (ms_stream * 1e-3), 1),
                                 "bound": "valu (64 pair tests per wave step); each 64-point tile is reused for 8 centres",
                                 "traffic": _pmc_traffic("ball_query_kernel")},
            "fps_sa1": {"ms_per_launch": round(fps_ms, 4), "point_updates_per_s": round(
                b * 2047 * n / (fps_ms * 1e-3), 1), "us_per_iteration": round(fps_ms * 1e3 / 2047, 4)}}


def matcher_at_detection_size(batch):
    """butd_hungarian_match at the detection split's size: 7 prefixes x B scenes, 66..132 targets x 256 queries
    (the timed step's synthetic targets have 1..16 boxes per scene, like the grounding splits)."""
    from butd_detr_amd import losses
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device=dev).manual_seed(0)
    P, Q, G = 7, 256, 132
    cost = torch.rand(P * batch, G, Q, device=dev, generator=g)      # (problems, targets, queries)
    n_valid = torch.randint(66, 133, (P * batch,), device=dev, generator=g)
    valid = torch.arange(G, device=dev)[None, :] < n_valid[:, None]
    stream = torch.cuda.current_stream()
    for _ in range(3):
        losses.hungarian_match(cost, valid)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for s, e in evs:
        s.record(stream)
        losses.hungarian_match(cost, valid)
        e.record(stream)
    torch.cuda.synchronize()
    ms = sum(s.elapsed_time(e) for s, e in evs) / len(evs)
    return {"kernel": "lsap_kernel (butd_hungarian_match)", "problems": P * batch, "queries": Q,
            "targets_per_problem": "66..132", "avg_launch_ms": round(ms, 4)}


def cpu_baseline(args, scenes):
    """Reference-shaped CPU path: this repo's host modules + the oracle's C operators (the reference
    has no CPU operators of its own) + torch CPU fp32, all host cores; bounded sample."""
    from butd_detr_amd import attention_blocks, pointnet2_utils
    from butd_detr_amd.train_step import make_optimizer, synthetic_batch, train_step
    from oracle import ext_adapter, pointnet2_oracle
    cores = pointnet2_oracle.host_cores()   # affinity capped by the cgroup quota (16 on the GPU boxes)
    torch.set_num_threads(cores)
    pointnet2_oracle.set_num_threads(cores)
    prev_ext, prev_backend = pointnet2_utils._ext, attention_blocks.get_backend()
    pointnet2_utils._ext = ext_adapter          # cpu_baseline leg only: oracle as the timed CPU port
    attention_blocks.set_backend("torch")
    try:
        ns = argparse.Namespace(**vars(args))
        ns.backend = "torch"
        model, _ = build_model(ns, torch.device("cpu"))
        opt = make_optimizer(model)
        inputs, targets = synthetic_batch(scenes, torch.device("cpu"), n_points=args.points,
                                          tokens=args.tokens)
        criterion = make_criterion(args)
        if criterion is not None:   # cpu_baseline leg only: scipy on the host, as the reference does it
            from oracle import lsap_oracle
            criterion.set_criterion.matcher.match_dense = lsap_oracle.scipy_match_dense(criterion.set_criterion.matcher)
        train_step(model, opt, inputs, targets, criterion=criterion)      # warm-up
        t0 = time.perf_counter()
        reps = 3 if scenes <= 2 else 2          # (bounded sample: ~1.2 s per scene and step on 16 cores)
        for _ in range(reps):
            train_step(model, opt, inputs, targets, criterion=criterion)
        dt = (time.perf_counter() - t0) / reps
    finally:
        pointnet2_utils._ext = prev_ext
        attention_blocks.set_backend(prev_backend)
    return {"value": round(scenes / dt, 4), "unit": "scenes/s", "cores": cores, "kind": "port",
            "sample": f"{reps} timed fwd+bwd+optimizer steps of {scenes} scene(s) x {args.points} points "
                      f"after 1 warm-up, oracle C ops (OpenMP) + torch CPU fp32"
                      + (", reference criterion with scipy's linear_sum_assignment" if criterion is not None else "")}


def _time_recaptured(args, model, opt, criterion, batches, warm=None):
    """The timed loop of main() on a freshly captured step (for the extra operating points) -> (seconds, loss)."""
    from butd_detr_amd.train_step import GraphedTrainStep
    graphed = GraphedTrainStep(model, opt, criterion=criterion)
    n, warm = len(batches), max(args.warmup, 1) if warm is None else warm
    for k in range(warm):
        graphed(*batches[k % n], next_inputs=batches[(k + 1) % n][0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(warm, warm + args.steps):
        loss = graphed(*batches[k % n], next_inputs=batches[(k + 1) % n][0])
    torch.cuda.synchronize()
    return time.perf_counter() - t0, loss


def text_cache_row(args, model, opt, criterion, batches):
    """SURVEY section 8(f)-3 as an EXTRA operating point: the frozen language model's hidden states per utterance
    kept in HBM (text_stream.UtteranceCache, opt-in while training because it removes the frozen tower's dropout
    noise) -- after the first pass over the batches RoBERTa no longer runs.  Not the headline: the reference
    encodes every batch."""
    from butd_detr_amd import text_stream
    model.text_cache = text_stream.UtteranceCache(next(model.parameters()).device, cache_in_training=True)
    try:
        dt, loss = _time_recaptured(args, model, opt, criterion, batches, warm=max(args.warmup, len(batches) + 1))
        c = model.text_cache
        return {"value": round(args.batch * args.steps / dt, 3), "unit": "scenes/s",
                "ms_per_step": round(dt / args.steps * 1e3, 3), "utterances": len(c), "store_bytes": c.bytes(),
                "batches_served_from_hbm": c.hits, "batches_encoded": c.misses, "final_loss": round(float(loss), 4),
                "note": "frozen RoBERTa replaced by one row gather per batch once its utterances are resident; "
                        "the text tower then behaves as in eval() (no dropout inside it)"}
    finally:
        model.text_cache = None


def bf16_row(args, model, opt, criterion, batches):
    """BASELINE configs[3]'s arithmetic as an EXTRA operating point (the headline stays the reference's fp32):
    the same step re-captured with the grouped products on the bf16 matrix cores, timed the same way."""
    from butd_detr_amd import fused_attention
    fused_attention.set_compute_dtype("bf16")
    model.text_precision = "bf16"
    try:
        inputs, targets = batches[0]
        dt, loss = _time_recaptured(args, model, opt, criterion, batches)
        ns = argparse.Namespace(**vars(args))
        ns.dtype = "bf16"
        gemm = gemm_roofline(ns)
        gemm["peak"] = BF16_MATRIX_PEAK_TF
        gemm["frac"] = round(gemm["achieved"] / BF16_MATRIX_PEAK_TF, 4)
        gemm["traffic"] = None
        gemm["kernel"] = gemm["kernel"].replace("fp32 MFMA", "bf16 MFMA (v_mfma_f32_16x16x32_bf16)")
        gemm["note"] = ("at the bf16 matrix rate these products are bound by staging their fp32 operands "
                        "(HBM / L2 -> LDS), not by the matrix pipe")
        return {"value": round(args.batch * args.steps / dt, 3), "unit": "scenes/s",
                "ms_per_step": round(dt / args.steps * 1e3, 3),
                "dtype": "bf16 operands / f32 accumulate in every grouped product, in the attention core's matrix "
                         "steps and in the frozen language model's linear layers; statistics, index ops and all "
                         "tensors in memory f32",
                "final_loss": round(float(loss), 4), "roofline": gemm,
                "roofline_attention": attention_roofline(args.batch, bf16=True)}
    finally:
        fused_attention.set_compute_dtype("f32")
        model.text_precision = "f32"


def encoder6_row(args, device, criterion, batches):
    """BASELINE configs[2] words the encoder as a "6-layer BiEncoder" (the reference hard-codes 3, bdetr.py:104):
    the same step with num_encoder_layers = 6 as an EXTRA operating point."""
    from butd_detr_amd.train_step import FlatAdamW
    ns = argparse.Namespace(**vars(args))
    ns.encoder_layers = 6
    model, _ = build_model(ns, device)
    dt, loss = _time_recaptured(args, model, FlatAdamW(model), criterion, batches)
    return {"value": round(args.batch * args.steps / dt, 3), "unit": "scenes/s",
            "ms_per_step": round(dt / args.steps * 1e3, 3), "encoder_layers": 6, "decoder_layers": 6,
            "final_loss": round(float(loss), 4)}


def torch_backend_row(args, device, criterion, batches):
    """What a straightforward PyTorch-ROCm port runs at on the same GPU, as an EXTRA row: the same model, criterion, batches and
    hipGraph step with ``--backend torch`` -- every attention / FFN / set-abstraction / Conv+BN block as stock torch ops
    (rocBLAS / hipBLASLt / MIOpen / ATen kernels), only the nine index ops of pointnet2._ext (which have no stock
    counterpart) and the device-side assignment from this library.  The closest thing to "the reference on MI355X" that
    can run here: the reference's own CUDA extension cannot be built."""
    from butd_detr_amd import attention_blocks
    from butd_detr_amd.train_step import FlatAdamW
    prev_backend, prev_strict = attention_blocks.get_backend(), attention_blocks.set_strict(False)
    ns = argparse.Namespace(**vars(args))
    ns.backend = "torch"
    ns.steps = max(4, args.steps // 3)
    try:
        model, _ = build_model(ns, device)                 # (sets the backend to "torch")
        dt, loss = _time_recaptured(ns, model, FlatAdamW(model), criterion, batches, warm=3)
    finally:
        attention_blocks.set_backend(prev_backend)
        attention_blocks.set_strict(prev_strict)
    return {"value": round(args.batch * ns.steps / dt, 3), "unit": "scenes/s", "ms_per_step": round(dt / ns.steps * 1e3, 3),
            "attention_backend": "torch", "steps": ns.steps, "final_loss": round(float(loss), 4),
            "note": "stock torch ops for every block + this library's index ops and assignment, same hipGraph step"}


def in_situ_row(args, model, opt, criterion, batches, gemm, sa_lin):
    """What the kernels achieve INSIDE the step (the stand-alone microbenchmarks of one site are the best case):
    * attention core over ALL sites of the step, from the roofline child's trace: the flops of every attention call
      (4 B H Lq Lk D forward, 2.5 x backward; counted in the child) / the summed durations of the attention kernels;
    * per-region time and matrix-core rate from marks captured into the step's graph (butd_detr_amd/step_regions.py:
      s_memrealtime stamps at region boundaries, no profiler; the marks themselves cost ~0.2 ms per step).  Region flops
      = grouped-GEMM problems + attention calls launched inside the region; the backbone's products that left the
      grouped GEMM (roofline_sa_linear.matrix_flops_per_step) are added to the backbone."""
    from butd_detr_amd import step_regions
    from butd_detr_amd.train_step import GraphedTrainStep
    out = {}
    steps = gemm.get("steps_in_trace") or 0
    if _LAST_CHILD_ROWS and steps and gemm.get("attention_flops_per_step"):
        names = ("attn_fwd_kernel", "attn_bwd", "attn_dq_fold")      # ("attn_fwd" alone is the language model's SDPA kernel)
        per = {}
        for n, _, t in _LAST_CHILD_ROWS:
            for key in names:
                if key in n:
                    per[key] = per.get(key, 0.0) + t / steps * 1e-6
        ms = sum(per.values())
        tf = gemm["attention_flops_per_step"] / (ms * 1e-3) / 1e12
        out["attention_core"] = {
            "what": "every attention call of one training step (%d forward + backward launches' worth of sites: 1024x1024, "
                    "80x80, 80x1024, 1024x80, 1024x132 in the encoder; 256x256, 256x80, 256x132, 256x1024 in the decoder)"
                    % gemm["attention_calls_per_step"],
            "gflop_per_step": round(gemm["attention_flops_per_step"] * 1e-9, 2), "ms_per_step": round(ms, 3),
            "ms_per_kernel_family": {k: round(v, 3) for k, v in per.items()},
            "achieved": round(tf, 2), "peak": FP32_MATRIX_PEAK_TF, "unit": "TFLOP/s",
            "frac": round(tf / FP32_MATRIX_PEAK_TF, 4),
            "source": "rocprofv3 --kernel-trace of the captured step (the roofline child)"}
        fam = gemm["algorithmic_flops_per_step"] + gemm["attention_flops_per_step"]
        fam_ms = gemm["ms_per_step_in_kernel"] + ms
        out["attention_plus_products"] = {
            "what": "gemm_kernel + the attention kernels together (projections, FFN, heads, set-abstraction products on "
                    "the grouped GEMM, and the attention core)",
            "ms_per_step": round(fam_ms, 3), "achieved": round(fam / (fam_ms * 1e-3) / 1e12, 2),
            "frac": round(fam / (fam_ms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TF, 4)}
    graphed = GraphedTrainStep(model, opt, criterion=criterion)
    marks = step_regions.StepMarks(model, graphed, next(model.parameters()).device).install()
    try:
        n = len(batches)
        for k in range(max(args.warmup, 4)):
            graphed(*batches[k % n], next_inputs=batches[(k + 1) % n][0])
        res = marks.measure(batches, replays=12)
    finally:
        marks.remove()
    regions = res["regions"]
    if sa_lin and "backbone" in regions:           # the products of csrc/sa_last_bwd.hip are not grouped-GEMM calls
        r = regions["backbone"]
        r["gflop"] = round(r["gflop"] + sa_lin["matrix_flops_per_step"] * 1e-9, 2)
        r["tflops"] = round(r["gflop"] / r["ms"], 2)
        r["frac_of_fp32_matrix_peak"] = round(r["tflops"] / FP32_MATRIX_PEAK_TF, 4)
    out["regions"] = regions
    out["regions_ms_per_step_with_marks"] = round(res["ms_per_step"], 3)
    out["regions_marks"] = res["marks"]
    out["regions_source"] = ("s_memrealtime marks captured into the step's hipGraph (butd_detr_amd/step_regions.py), mean of "
                             "10 free-running replays; forward intervals belong to the module whose mark ends them, backward "
                             "intervals to the module whose mark starts them")
    if os.environ.get("BUTD_STEP_MARKS_OUT"):
        with open(os.environ["BUTD_STEP_MARKS_OUT"], "w") as f:
            f.write(step_regions.format_intervals(res) + "\n")
    return out


def batch24_row(args, model, opt, criterion, device):
    """The reference's own training batch (scripts/train_test_det.sh:6: --batch_size 24) as an EXTRA operating
    point: the same step at 24 scenes per GPU -- how much of the headline's matrix-core fraction is the 2048-row
    granularity of B = 8 (the decoder's products have 24 x 256 = 6144 rows here)."""
    from butd_detr_amd.train_step import synthetic_batch
    B = 24
    batches = [synthetic_batch(B, device, seed=2211 + 37 * k, n_points=args.points, tokens=args.tokens,
                               max_targets=args.max_targets) for k in range(2)]
    ns = argparse.Namespace(**vars(args))
    ns.steps = max(4, args.steps // 3)
    dt, loss = _time_recaptured(ns, model, opt, criterion, batches, warm=3)
    return {"value": round(B * ns.steps / dt, 3), "unit": "scenes/s", "ms_per_step": round(dt / ns.steps * 1e3, 3),
            "scenes_per_step": B, "ms_per_scene": round(dt / ns.steps * 1e3 / B, 3), "steps": ns.steps,
            "final_loss": round(float(loss), 4)}


def _flush_c_stdio():
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def make_criterion(args):
    if args.criterion == "surrogate":
        return None
    from butd_detr_amd.train_step import HungarianCriterion
    return HungarianCriterion(num_decoder_layers=6, use_contrastive_align=True, use_soft_token_loss=True)


def _spawn_ranks(args):
    """``python bench.py --gpus N`` without a launcher: start the N ranks ourselves, exactly the way the driver's
    torch.distributed.run line does (one process per GPU, env:// rendezvous on 127.0.0.1, RCCL), and hand its exit code
    back.  The ranks' stdout is inherited, so rank 0's JSON line is this command's last line too."""
    import socket
    import subprocess
    if not os.environ.get("BUTD_BENCH_ONE_GPU") and torch.cuda.device_count() < args.gpus:
        sys.exit(f"bench.py --gpus {args.gpus}: this node shows {torch.cuda.device_count()} GPU(s)")
    with socket.socket() as sock:              # a free rendezvous port (released before torchrun binds it)
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.roofline_child:
        _spawn_ranks(args)                     # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and not args.roofline_child and os.environ.get("BUTD_BENCH_FORCE_DIST") != "1":
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    # BUTD_BENCH_FORCE_DIST=1: bring RCCL up at world size 1 and issue the step's collectives anyway (what a
    # one-GPU box can check of the N > 1 path: RCCL next to hipGraph replays; tests/test_gpu_two_ranks.py)
    force_dist = os.environ.get("BUTD_BENCH_FORCE_DIST") == "1"
    if force_dist:
        os.environ["BUTD_FORCE_COLLECTIVE"] = "1"
        for k, v in (("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # backend "nccl" IS RCCL on ROCm.  BUTD_BENCH_BACKEND=gloo + BUTD_BENCH_ONE_GPU=1 exist only so that the
        # N>1 code path can be rehearsed on a one-GPU box (tests/test_gpu_two_ranks.py): RCCL refuses two ranks
        # on one device
        dist.init_process_group(backend=os.environ.get("BUTD_BENCH_BACKEND", "nccl"), init_method="env://")
    ranks_in_group = dist.get_world_size() if dist.is_initialized() else 1     # what the backend itself reports
    assert ranks_in_group == world, (ranks_in_group, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    device = torch.device("cuda", 0 if os.environ.get("BUTD_BENCH_ONE_GPU") else local_rank)
    torch.cuda.set_device(device)

    from butd_detr_amd.train_step import (GraphedTrainStep, make_optimizer, synthetic_batch,
                                          train_step as eager_step, wrap_data_parallel)
    model, backend = build_model(args, device)
    if backend == "hip":
        from butd_detr_amd import fused_attention
        fused_attention.set_compute_dtype(args.dtype)
    # the timed steps rotate over several batches: every step sees other scenes than the one before it
    batches = [synthetic_batch(args.batch, device, seed=1184 + 37 * k, n_points=args.points, tokens=args.tokens,
                               rank=rank, max_targets=args.max_targets)
               for k in range(max(1, args.distinct_batches))]
    inputs, targets = batches[0]
    criterion = make_criterion(args)
    # a copy of the targets that already carries the rank-averaged box count, for the rank-0-only eager step of
    # the roofline section (criterion.prepare is a collective: every rank calls it here)
    local_targets = criterion.prepare(targets) if criterion is not None else targets
    if args.eager:
        ddp = wrap_data_parallel(model, device)
        opt = make_optimizer(model)

        def train_step(k):
            i, t = batches[k % len(batches)]
            return eager_step(ddp, opt, i, t, criterion=criterion)
    else:
        from butd_detr_amd.train_step import FlatAdamW
        opt = FlatAdamW(model)
        graphed = GraphedTrainStep(model, opt, criterion=criterion, overlap_exchange=args.overlap_exchange)
        ddp = model

        def train_step(k):       # batch k now; batch k+1 announced (its FPS chain / text run under this step)
            i, t = batches[k % len(batches)]
            return graphed(i, t, next_inputs=batches[(k + 1) % len(batches)][0])

    if args.roofline_child:
        # under rocprofv3 (parent: gemm_roofline): the captured step a few times, then the GEMM work of one step
        for k in range(args.warmup + args.steps):
            train_step(k)
        torch.cuda.synchronize()
        work = gemm_work(lambda: eager_step(model, make_optimizer(model), inputs, local_targets, criterion=criterion))
        _flush_c_stdio()
        print(json.dumps(work), flush=True)
        return
    for k in range(args.warmup):
        train_step(k)
    _flush_c_stdio()            # RCCL's start-up banner sits in the C stdio buffer of every rank: out with it now, so
                                # that the JSON line below is the LAST line on stdout
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.warmup, args.warmup + args.steps):
        loss = train_step(k)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        scenes = args.batch * world * args.steps
        out = {
            "metric": "scenes/sec fwd+bwd (50k pts, 256 queries, 80 tokens)",
            "value": round(scenes / elapsed, 3), "unit": "scenes/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.dtype == "f32" else "bf16 operands / f32 accumulate (grouped products, attention "
                                                       "matrix steps); f32 elsewhere",
            "data": f"synthetic ({len(batches)} different batches per rank, rotated)",
            "config": {"workload": f"BASELINE configs[2]/[3]: {args.batch} scenes/GPU x {args.points} "
                                   f"points, {args.queries} queries, {args.tokens} tokens, 132 box slots, "
                                   f"{args.encoder_layers} encoder + 6 decoder layers, 1..{args.max_targets} "
                                   "targets per scene, fwd+loss+bwd+clip+AdamW, train mode",
                       "criterion": ("compute_hungarian_loss (matcher 1/0/2, soft token + contrastive align, "
                                     "assignment on the device)" if criterion is not None else "dense surrogate"),
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "ranks_in_process_group": ranks_in_group,
                       "collective_backend": (dist.get_backend() if dist.is_initialized() else None),
                       "attention_backend": backend, "launch": "eager+DDP+torch AdamW" if args.eager else "hipGraph replay (FPS chain of the next batch prefetched on a forked stream) + flat-gradient all-reduce + packed AdamW", "final_loss": round(float(loss), 4)},
        }
        if backend == "hip":
            # rank 0 only from here on: no collective may run (the criterion's box count was all-reduced above)
            out["roofline"] = gemm_roofline(args, lambda: eager_step(model, make_optimizer(model), inputs, local_targets,
                                                                         criterion=criterion))
            sa_lin = sa_linear_roofline(args, out["roofline"]) if args.dtype == "f32" else None
            if sa_lin:
                out["roofline_sa_linear"] = sa_lin
            if world == 1 and not args.eager and args.dtype == "f32" and os.environ.get("BUTD_BENCH_NO_CHILD") != "1":
                out["in_situ"] = in_situ_row(args, model, opt, criterion, batches, out["roofline"], sa_lin)
            if not args.no_extras:
                out["roofline_ball_query"] = ball_query_roofline(inputs)
                out["roofline_attention"] = attention_roofline(args.batch)
                out["roofline_attention"]["scope"] = ("stand-alone, best site (the encoder's 1024 x 1024 visual self-attention); "
                                                      "in_situ.attention_core is the figure over all sites of the step")
                out["matcher_detection_split"] = matcher_at_detection_size(args.batch)
        else:
            out["roofline"] = ball_query_roofline(inputs)
        extras = (backend == "hip" and world == 1 and args.dtype == "f32" and not args.eager and not args.no_extras)
        if extras and not args.no_bf16_row:
            out["bf16_operating_point"] = bf16_row(args, model, opt, criterion, batches)
        if extras:
            out["text_cache_operating_point"] = text_cache_row(args, model, opt, criterion, batches)
            if args.encoder_layers != 6:
                out["encoder6_operating_point"] = encoder6_row(args, device, criterion, batches)
            out["batch24_operating_point"] = batch24_row(args, model, opt, criterion, device)
            out["torch_backend_operating_point"] = torch_backend_row(args, device, criterion, batches)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_scenes)
        _flush_c_stdio()
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
