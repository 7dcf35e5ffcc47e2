"""hipGraph hygiene of the captured training step (include/butd_graph.h).

Why it exists: on ROCm 7.2 a MEMSET node of a replayed hipGraph is unreliable --
replayed behind a still-running graph (any size), or simply a second time (multi-megabyte ranges), it writes a
garbage pattern instead of its value (scratch/graph_node_order.py; correct with
``DEBUG_CLR_GRAPH_PACKET_CAPTURE=0``).  Every captured ``hipMemsetAsync`` is affected, e.g. the semaphore reset of
torch's multi-block reductions (ATen/native/cuda/Reduce.cuh): ``vector_norm`` of the packed gradient buffer then
never wrote its result, the clip coefficient became 1.0 and the update was applied unclipped -- round 2's
"free-running divergence".  The training step therefore (i) issues no hipMemsetAsync of its own (csrc/zero_fill.h),
(ii) rewrites the memset nodes torch ops leave in a captured graph into kernel nodes (``make_safe``), and
(iii) checks that none is left (``inventory``).
"""
import collections
import ctypes

from . import _hiplib

NODE_TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event",
              7: "event_record", 8: "ext_sem_signal", 9: "ext_sem_wait", 10: "mem_alloc", 11: "mem_free",
              12: "memcpy_from_symbol", 13: "memcpy_to_symbol"}


# where the memset-node bug was observed (round 3, profiles/r03_hipgraph_memset_nodes.txt: 200 of 200 replays wrong; round 4,
# profiles/r04_runtime_probe.json: 100 of 100): the ROCm 7.2.0 image with torch 2.10.0+rocm7.0 on MI355X (gfx950); the
# process runs torch's BUNDLED HIP runtime (hipRuntimeGetVersion 70051831 = 7.0.51831), not /opt/rocm's.
# tests/test_gpu_runtime_probe.py replays the reproducer, reports whether the running stack still has the bug and asserts
# that the rewritten graph is right either way (once a runtime fixes it the rewrite is merely redundant).
BUG_SEEN_ON = {"image": "ROCm 7.2.0", "hip_runtime_in_process": 70051831, "torch": "2.10.0+rocm7.0", "arch": "gfx950"}


def runtime_versions():
    """{'hip_runtime': int, 'hip_driver': int, 'torch': str, 'torch_hip': str} of this process."""
    import torch
    rt, drv = ctypes.c_int(0), ctypes.c_int(0)
    _hiplib.check(_hiplib.load().butd_runtime_versions(ctypes.byref(rt), ctypes.byref(drv)), "butd_runtime_versions")
    return {"hip_runtime": rt.value, "hip_driver": drv.value, "torch": torch.__version__,
            "torch_hip": str(torch.version.hip)}


def new_graph():
    """A ``torch.cuda.CUDAGraph`` whose hipGraph_t stays accessible after capture (instantiated at first replay)."""
    import torch
    need = ("raw_cuda_graph", "instantiate")
    if any(not hasattr(torch.cuda.CUDAGraph, n) for n in need):
        raise RuntimeError(f"graph_audit.new_graph: torch {torch.__version__} lacks CUDAGraph(keep_graph=True) / "
                           f"{' / '.join(need)} (torch >= 2.8 has them): the captured step cannot be scrubbed of "
                           "memset nodes on this torch, and replaying them is unsafe on ROCm 7.2")
    return torch.cuda.CUDAGraph(keep_graph=True)


def inventory(cuda_graph):
    """Counter of node kinds of a captured ``new_graph()``."""
    counts = (ctypes.c_int * 16)()
    _hiplib.check(_hiplib.load().butd_graph_node_counts(cuda_graph.raw_cuda_graph(), ctypes.byref(counts)),
                  "butd_graph_node_counts")
    return collections.Counter({NODE_TYPES.get(i, f"type{i}"): c for i, c in enumerate(counts) if c})


def make_safe(cuda_graph):
    """Rewrite the memset nodes of a captured, not yet replayed ``new_graph()`` into kernel nodes; returns how many
    were rewritten and raises if any memset node is left."""
    n = ctypes.c_int(0)
    _hiplib.check(_hiplib.load().butd_graph_replace_memset_nodes(cuda_graph.raw_cuda_graph(), ctypes.byref(n)),
                  "butd_graph_replace_memset_nodes")
    left = inventory(cuda_graph).get("memset", 0)
    if left:
        raise RuntimeError(f"{left} memset node(s) left in a captured graph: replaying it is unsafe on this ROCm "
                           "(butd_detr_amd/graph_audit.py)")
    return n.value


_OWN_STREAMS = {}
_OWNED, _FREE = {}, {}      # (device, role, id(owner)) -> stream;  (device, role) -> streams whose owner is gone


def _create_stream(dev, priority):
    import torch
    handle = ctypes.c_void_p(0)
    with torch.cuda.device(dev):
        _hiplib.check(_hiplib.load().butd_stream_create(int(priority), ctypes.byref(handle)), "butd_stream_create")
        return torch.cuda.ExternalStream(handle.value, device=dev)


def _release(key):
    stream = _OWNED.pop(key, None)
    if stream is not None:
        _FREE.setdefault(key[:2], []).append(stream)


def own_stream(device=None, role="side", priority=0, owner=None):
    """A stream of our own (``butd_stream_create``) wrapped as ``torch.cuda.ExternalStream`` -- NOT a member of torch's
    round-robin pool of 32, which RCCL's stream and torch's default capture stream come from too (include/butd_graph.h).
    Everything the captured step forks, captures on or uploads through is created here.  One stream per (device, role)
    and process, never destroyed: torch's caching allocator keeps per-stream state keyed by the raw handle (block
    pools, record_stream events), so a destroyed handle would be touched again by a later empty_cache().
    ``owner`` (any weak-referenceable object: a GraphedTrainStep, a model): the stream belongs to that instance -- two steps
    or models of one process (train + eval, the two-model tests) do not serialise their prefetch branches behind each
    other, and a capture of one does not put a stream into capture mode on which the other still has eager work.  When the
    owner is collected its streams go to a free list and serve the next owner of that role (still never destroyed)."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    if owner is not None:
        import weakref
        key = (dev.index, role, id(owner))
        stream = _OWNED.get(key)
        if stream is None:
            free = _FREE.get(key[:2])
            stream = free.pop() if free else _create_stream(dev, priority)
            _OWNED[key] = stream
            weakref.finalize(owner, _release, key)
        return stream
    key = (dev.index, role)
    stream = _OWN_STREAMS.get(key)
    if stream is None:
        stream = _OWN_STREAMS[key] = _create_stream(dev, priority)
    return stream
