"""-m gpu: the captured training step in the regime ``bench.py`` times -- the host never reads anything back, so it
enqueues step k+1 while step k still runs.

Round 2 left an "open problem": free-running, the step trained measurably worse although every replay, checked
alone, was right.  Cause: a MEMSET node of a hipGraph replayed behind a still-running graph
writes garbage on ROCm 7.2; torch's ``vector_norm`` of the packed gradients zeroes a semaphore with such a node,
then never wrote its result, the clip coefficient became 1.0 and the update went out unclipped.  Pinned here:

* no memset node survives in any captured graph of the step (they are rewritten into kernel nodes);
* with learning rate 0 and the dropout counter pinned per batch EVERY free-running replay reproduces the first
  visit of its batch -- loss, clip coefficient and a checksum of the packed gradients, logged on the device;
* at the reference's learning rates 60 free-running steps over 3 batches end where the same run with a host
  wait before every step and the eager packed-optimizer run end (fp32 atomics make trajectories chaotic at the
  1 % level -- measured spread of identical runs: +-1.6 % -- so the bound is 5 % on the mean of the last
  two visits of every batch; an unclipped update moved it by 8-18 %),
for {single graph, two-piece overlapped exchange} x {fan_out on, off} x {prefetch branches on, off}."""
import gc
import os
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu

STEPS, NB = 60, 3
CONFIGS = [(split, fan, pre) for split in (False, True) for fan in ("1", "0") for pre in (True, False)]


def _model():
    from butd_detr_amd import attention_blocks
    from butd_detr_amd.bdetr import BeaUTyDETR
    from butd_detr_amd.offline_text import offline_factory
    attention_blocks.set_backend("hip")
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = BeaUTyDETR(num_class=256, num_obj_class=485, input_feature_dim=3, num_queries=256,
                       num_decoder_layers=6, self_position_embedding="loc_learned", contrastive_align_loss=True,
                       butd=True, self_attend=True, text_encoder_factory=offline_factory(0), num_encoder_layers=3)
    m = m.cuda().train()
    m.text_encoder.eval()                         # stock Philox dropouts off: runs must be comparable
    for mod in m.text_projector.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m


@pytest.fixture(scope="module")
def batches():
    from butd_detr_amd.train_step import synthetic_batch
    dev = torch.device("cuda", 0)
    return [synthetic_batch(8, dev, seed=1184 + 50 * i, n_points=50000, tokens=80) for i in range(NB)]


@pytest.fixture(scope="module")
def process_group():
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group("nccl", init_method="env://", rank=0, world_size=1)   # RCCL next to the replays
    yield
    dist.destroy_process_group()


def _pinned_counter():
    from butd_detr_amd import fused_attention as fa
    return fa.rng_counter(torch.device("cuda", 0))


@pytest.fixture(scope="module")
def eager_tail(batches, process_group):
    """Mean loss of the last two visits of every batch of the EAGER run with the packed optimizer."""
    from butd_detr_amd.train_step import FlatAdamW, HungarianCriterion
    model = _model()
    opt, crit = FlatAdamW(model), HungarianCriterion()
    ctr, log = _pinned_counter(), torch.zeros(STEPS, device="cuda")
    for it in range(STEPS):
        inp, tgt = batches[it % NB]
        ctr.fill_(5000 + it)
        loss = crit(model(inp), crit.prepare(tgt))
        opt.zero_grad()
        loss.backward()
        opt.collect_grads()
        opt.clip_(0.1)
        opt.step(packed=True)
        log[it].copy_(loss.detach())
    tail = float(log[-2 * NB:].mean())
    del model, opt
    gc.collect()
    torch.cuda.empty_cache()
    return tail


def _run(step, batches, n, counter_base, pin_per_batch):
    """n free-running steps; (loss, clip coefficient, gradient checksum) per step, logged on the device."""
    opt, ctr = step.optimizer, _pinned_counter()
    log = torch.zeros(n, 3, device="cuda")
    for it in range(n):
        inp, tgt = batches[it % NB]
        ctr.fill_(counter_base + (it % NB if pin_per_batch else it))
        loss = step(inp, tgt, next_inputs=batches[(it + 1) % NB][0])
        log[it, 0].copy_(loss)
        log[it, 1].copy_(opt.grad_scale[0])
        log[it, 2].copy_(opt.flat_g[::97].abs().sum())
    torch.cuda.synchronize()
    return log.cpu().double()


@pytest.mark.parametrize("split,fan_out,prefetch", CONFIGS,
                         ids=[f"{'two_piece' if s else 'single'}-fanout{f}-{'prefetch' if p else 'inline'}"
                              for s, f, p in CONFIGS])
def test_free_running_step(split, fan_out, prefetch, batches, process_group, eager_tail):
    from butd_detr_amd import attention_blocks
    from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, HungarianCriterion
    from butd_detr_amd import fan_out as fan_out_mod
    env = {"BUTD_FORCE_COLLECTIVE": "1" if split else "0"}
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    prev_fan = fan_out_mod.set_enabled(fan_out == "1")
    model = step = None
    try:
        model = _model()
        opt = FlatAdamW(model)
        step = GraphedTrainStep(model, opt, criterion=HungarianCriterion(), prefetch_sampling=prefetch,
                                prefetch_text=prefetch, overlap_exchange=split)
        assert step.split == split
        _pinned_counter().fill_(1)
        step(batches[0][0], batches[0][1], next_inputs=batches[0][0])       # capture + one replay
        slot = step._slot
        for name, kinds in slot.node_inventory.items():
            assert kinds.get("memset", 0) == 0, (name, kinds, slot.memset_nodes_rewritten)
            assert kinds.get("kernel", 0) > 0
        torch.cuda.synchronize()
        snap = step._snapshot()
        lrs = [(g["lr"], g["weight_decay"]) for g in opt.param_groups]

        # (1) learning rate 0: every free-running replay reproduces the first visit of its batch
        for g in opt.param_groups:
            g["lr"], g["weight_decay"] = 0.0, 0.0
        log = _run(step, batches, STEPS, 1000, pin_per_batch=True)
        for j in range(NB):
            seq = log[j::NB]
            ref = seq[1]                                   # (visit 0 of batch 0 follows the capture call)
            rel = ((seq[1:] - ref).abs() / ref.abs().clamp_min(1e-12)).max(0).values
            assert float(rel[0]) <= 1e-5 and float(rel[1]) <= 1e-4 and float(rel[2]) <= 1e-4, (j, rel.tolist())
            assert 0 < float(ref[1]) < 0.1, "the clip coefficient of this model is ~2e-3, never 1"

        # (2) the reference's learning rates: free-running == host wait before every step == eager
        tails = {}
        for mode, sync in (("free", "0"), ("synced", "1")):
            step._restore(snap)
            for g, (lr, wd) in zip(opt.param_groups, lrs):
                g["lr"], g["weight_decay"] = lr, wd
            step.step_sync = sync == "1"
            torch.cuda.synchronize()
            tails[mode] = float(_run(step, batches, STEPS, 5000, pin_per_batch=False)[-2 * NB:, 0].mean())
        assert abs(tails["free"] - tails["synced"]) <= 0.05 * tails["synced"], (tails, eager_tail)
        assert abs(tails["free"] - eager_tail) <= 0.05 * eager_tail, (tails, eager_tail)
    finally:
        fan_out_mod.set_enabled(prev_fan)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        del step, model
        gc.collect()
        torch.cuda.empty_cache()
        attention_blocks.set_backend("torch")


def test_memset_nodes_of_a_captured_graph_become_kernel_nodes():
    """The rewrite itself: 1-, 2- and 4-byte patterns, unaligned ranges, a pitched 2-D memset and the semaphore
    reset of a multi-block torch reduction, replayed behind a still-running graph."""
    import ctypes
    from butd_detr_amd import graph_audit
    hip = ctypes.CDLL("libamdhip64.so")
    P = ctypes.c_void_p
    hip.hipMemsetAsync.argtypes = [P, ctypes.c_int, ctypes.c_size_t, P]
    hip.hipMemsetD16Async.argtypes = [P, ctypes.c_ushort, ctypes.c_size_t, P]
    hip.hipMemsetD32Async.argtypes = [P, ctypes.c_int, ctypes.c_size_t, P]
    hip.hipMemset2DAsync.argtypes = [P, ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, P]
    dev = torch.device("cuda", 0)
    a = torch.randn(1024, 1024, device=dev)
    flat = torch.randn(21_400_000, device=dev)
    raw = torch.zeros(1 << 16, dtype=torch.uint8, device=dev)
    want_sum, want_norm = float(flat.sum()), float(torch.linalg.vector_norm(flat))
    out = torch.zeros(2, device=dev)

    def busy():
        x = a
        for _ in range(30):
            x = torch.tanh(x @ a * 1e-2)
        return x

    def body():
        st = torch.cuda.current_stream().cuda_stream
        raw.fill_(7)
        base = raw.data_ptr()
        assert hip.hipMemsetAsync(base + 3, 0xAB, 1001, st) == 0                   # bytes [3, 1004)
        assert hip.hipMemsetD16Async(base + 2048 + 2, 0x1234, 77, st) == 0         # 77 halfwords from 2050
        assert hip.hipMemsetD32Async(base + 4096 + 4, 0x01020304, 33, st) == 0     # 33 words from 4100
        assert hip.hipMemset2DAsync(base + 8192 + 1, 100, 0x5C, 37, 9, st) == 0    # 9 rows of 37 bytes, pitch 100
        out[0].copy_(flat.sum())                                                   # semaphore memsets (torch)
        out[1].copy_(torch.linalg.vector_norm(flat))

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        busy(); body()
        torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            keep = busy()
        g2 = graph_audit.new_graph()
        with torch.cuda.graph(g2):
            body()
    before = graph_audit.inventory(g2)
    assert before.get("memset", 0) >= 5, before
    assert graph_audit.make_safe(g2) == before["memset"]
    after = graph_audit.inventory(g2)
    assert after.get("memset", 0) == 0 and after["kernel"] == before["kernel"] + before["memset"]
    want = torch.full((1 << 16,), 7, dtype=torch.uint8)
    want[3:1004] = 0xAB
    want[2050:2050 + 154] = torch.tensor([0x34, 0x12] * 77, dtype=torch.uint8)
    want[4100:4100 + 132] = torch.tensor([0x04, 0x03, 0x02, 0x01] * 33, dtype=torch.uint8)
    for r in range(9):
        want[8193 + 100 * r:8193 + 100 * r + 37] = 0x5C
    for _ in range(20):
        out.fill_(-1.0)
        g1.replay()
        g2.replay()                                          # behind a still-running graph, no host wait
    torch.cuda.synchronize()
    assert torch.equal(raw.cpu(), want)
    assert abs(float(out[0]) - want_sum) <= 1e-3 * abs(want_sum) + 1.0
    assert abs(float(out[1]) - want_norm) <= 1e-4 * want_norm
    del keep
