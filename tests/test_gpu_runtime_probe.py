"""-m gpu: is the ROCm runtime bug the captured step works around still present on THIS stack?

DESIGN.md section 7: on HIP runtime 7.2 (ROCm 7.2.0 image, torch 2.10.0+rocm7.0, MI355X) a MEMSET node of a
hipGraph replayed behind a still-running graph writes a garbage pattern instead of its value.  The step's
correctness rests on rewriting such nodes into kernel nodes (include/butd_graph.h).  This probe replays the
reproducer (scratch/graph_node_order.py, condensed) and

* records the runtime / driver / torch versions and whether the raw memset node misbehaves
  (``gpurun_out/runtime_probe.json``; printed with ``-s``) -- a REPORT, either answer passes: once a runtime
  fixes it the rewrite is merely redundant;
* asserts what the product relies on: after ``graph_audit.make_safe`` the same graph is right on every replay.
"""
import ctypes
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ITERS = 100


def _build(rewrite, explicit_instantiate=True):
    """scratch/graph_node_order.py, literally: [kernels; fill 1.0; MEMSET 4 MB -> 0; MEMSET 256 B -> 0; sums; add; MEMCPY;
    mean] replayed behind a still-running graph."""
    from butd_detr_amd import graph_audit
    dev = torch.device("cuda", 0)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    a = torch.randn(2048, 2048, device=dev)
    buf = torch.zeros(1 << 20, device=dev)           # 4 MB
    small = torch.zeros(64, device=dev)              # semaphore-sized
    A = torch.zeros(1 << 20, device=dev)
    B = torch.zeros(1 << 20, device=dev)
    out = torch.zeros(4, device=dev)

    def busy(k=40):
        x = a
        for _ in range(k):
            x = torch.tanh(x @ a * 1e-2)
        return x

    def body():
        st = torch.cuda.current_stream().cuda_stream
        keep = busy(4)
        buf.fill_(1.0)
        small.fill_(1.0)
        hip.hipMemsetAsync(buf.data_ptr(), 0, buf.numel() * 4, st)       # MEMSET nodes once captured
        hip.hipMemsetAsync(small.data_ptr(), 0, small.numel() * 4, st)
        out[0].copy_(buf[:4096].sum())
        out[1].copy_(small.sum())
        A.add_(1.0)
        B.copy_(A)                                                        # MEMCPY node
        out[2].copy_(B[:1024].mean())
        return keep

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            busy()
            body()
        torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            k1 = busy()
        g2 = graph_audit.new_graph()
        with torch.cuda.graph(g2):
            k2 = body()
    torch.cuda.synchronize()
    before = dict(graph_audit.inventory(g2))
    rewritten = graph_audit.make_safe(g2) if rewrite else 0
    if explicit_instantiate:
        g2.instantiate()          # (as GraphedTrainStep does after the scrub; otherwise the first replay instantiates)
    A.zero_()
    torch.cuda.synchronize()
    log = torch.zeros(ITERS, 4, device=dev)
    for i in range(ITERS):
        g1.replay()                                   # g2 is enqueued behind a still-running graph
        g2.replay()
        log[i].copy_(out)
    torch.cuda.synchronize()
    wrong = int((log[:, :2] != 0).any(dim=1).sum())
    return before, rewritten, wrong, (k1, k2)


def test_memset_node_bug_probe_and_rewrite():
    from butd_detr_amd import graph_audit
    versions = graph_audit.runtime_versions()
    before, _, wrong_raw, _ = _build(rewrite=False)
    assert before.get("memset", 0) == 2, before
    _, rewritten, wrong_fixed, _ = _build(rewrite=True)
    report = {"versions": versions, "bug_seen_on": graph_audit.BUG_SEEN_ON,
              "raw_memset_node_wrong_replays": wrong_raw, "replays": ITERS, "bug_present": wrong_raw > 0,
              "rewritten_nodes": rewritten,
              "rewritten_graph_wrong_replays": wrong_fixed}
    print("runtime probe:", json.dumps(report))
    root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "runtime_probe.json"), "w") as f:
            json.dump(report, f, indent=1)
    except OSError:
        pass
    assert rewritten == 2
    assert wrong_fixed == 0, report          # what the product relies on
