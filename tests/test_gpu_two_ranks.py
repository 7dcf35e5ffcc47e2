"""-m gpu: the bench's multi-rank step (hipGraph replay of forward+backward -> ONE all-reduce of the packed
gradient buffer -> hipGraph replay of clip + packed AdamW, with the sampling chain and the language model
prefetched) run by TWO processes.  The box has one GPU, so both ranks sit on cuda:0 and talk over gloo
(RCCL refuses two ranks on one device; the collective is the only thing that differs from the 8-GPU
launch, torch.distributed semantics are the same).  After every step the ranks must hold bit-identical
averaged gradients and parameters, although they train on different scenes."""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, init_file, out_dir, overlap):
    sys.path.insert(0, ROOT)
    import warnings
    from butd_detr_amd import attention_blocks
    from butd_detr_amd.bdetr import BeaUTyDETR
    from butd_detr_amd.offline_text import offline_factory
    from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, HungarianCriterion, synthetic_batch
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    attention_blocks.set_backend("hip")
    torch.manual_seed(0)                       # same initial weights on both ranks
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = BeaUTyDETR(num_class=256, num_obj_class=485, input_feature_dim=3, num_queries=32,
                           num_decoder_layers=1, num_encoder_layers=1, self_position_embedding="loc_learned",
                           contrastive_align_loss=True, butd=True, self_attend=True,
                           text_encoder_factory=offline_factory(0)).to(dev).train()
    opt = FlatAdamW(model)
    # the reference's criterion: its box count is averaged over the ranks (losses.py:527-534) outside the graph
    step = GraphedTrainStep(model, opt, warmup=1, criterion=HungarianCriterion(num_decoder_layers=1),
                            overlap_exchange=overlap)
    assert step.split == overlap
    batches = [synthetic_batch(2, dev, seed=11 + 5 * i, n_points=4096, tokens=16, rank=rank) for i in range(3)]
    losses = []
    for i, (inp, tgt) in enumerate(batches):
        nxt = batches[i + 1][0] if i + 1 < len(batches) else None
        losses.append(float(step(inp, tgt, next_inputs=nxt)))
    torch.cuda.synchronize()
    torch.save({"flat_p": opt.flat_p.cpu(), "flat_g": opt.flat_g.cpu(), "losses": losses},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True], ids=["one all-reduce", "two-piece overlapped exchange"])
def test_two_ranks_stay_in_lockstep(overlap):
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "init")
        mp.spawn(_worker, args=(2, init_file, tmp, overlap), nprocs=2, join=True)
        r0, r1 = (torch.load(os.path.join(tmp, f"r{r}.pt")) for r in (0, 1))
    assert r0["losses"] != r1["losses"]                     # different scenes per rank
    assert torch.equal(r0["flat_g"], r1["flat_g"])          # identical averaged gradients
    assert torch.equal(r0["flat_p"], r1["flat_p"])          # identical parameters after the updates
    assert torch.isfinite(r0["flat_p"]).all()


def test_bench_runs_its_multi_rank_path():
    """bench.py exactly as the driver launches it for N=2 (torch.distributed.run, env:// rendezvous on
    127.0.0.1, barriers, max-over-ranks timing, flat gradient all-reduce, one JSON line from rank 0) --
    rehearsed on one GPU over gloo, because RCCL refuses two ranks on one device."""
    import json
    import subprocess
    env = dict(os.environ, BUTD_BENCH_BACKEND="gloo", BUTD_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29653", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--points", "8192", "--queries", "64",
           "--tokens", "24"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["config"]["global_batch"] == 4
    assert rec["value"] > 0 and rec["scaling"] == "weak" and "cpu_baseline" not in rec


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """``python bench.py --gpus 2`` with NO torch.distributed.run in front and no WORLD_SIZE in the environment (the shape
    of the driver's N = 1 command with another N): bench.py must start the two ranks itself and report what the process
    group -- not the flag -- says.  Round 4's bench parsed --gpus and never read it."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BUTD_BENCH_BACKEND="gloo", BUTD_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
           "--points", "8192", "--queries", "64", "--tokens", "24"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["ranks_in_process_group"] == 2
    assert rec["config"]["global_batch"] == 4 and rec["config"]["collective_backend"] == "gloo"
    # a launcher that started another number of ranks than --gpus says is an error, not a silently different run
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], cwd=ROOT,
                         env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr


def test_bench_brings_rccl_up_next_to_the_graph_replays():
    """What one GPU can check of the N > 1 launch: ``backend="nccl"`` (RCCL) initialised at world size 1, the
    step's collectives issued anyway (BUTD_BENCH_FORCE_DIST=1 -> two-piece capture, asynchronous all-reduce of
    the decoder-side bucket under the replay of the encoder / backbone backward), HSA_ENABLE_IPC_MODE_LEGACY=0."""
    import json
    import subprocess
    env = dict(os.environ, BUTD_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--batch", "2", "--points", "8192", "--no-cpu-baseline"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and rec["config"]["final_loss"] == rec["config"]["final_loss"]
