"""Root cause probe for DESIGN.md 7.4 #6b (round 4: RCCL's watchdog thread aborted the process twice during a two-piece
capture, 'operation not permitted on an event last recorded in a capturing stream', in one hand-picked test order only).

Hypothesis: torch.cuda.Stream() is a member of a round-robin pool of 32 streams per priority; ProcessGroupNCCL takes ITS
stream from the same pool.  After enough Stream() calls in a process a 'new' side stream of the step IS RCCL's stream.
Forked inside a capture it puts RCCL's stream into capture mode; the watchdog's next poll of a finished collective's end
event (recorded on that stream) then fails with hipErrorCapturedEvent and the watchdog terminates the process.

    python scratch/stream_alias_probe.py pool     # side streams from torch's pool: expected to abort at one offset
    python scratch/stream_alias_probe.py own      # side streams from graph_audit.own_stream: expected to pass all 32
"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "pool"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29655")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
x = torch.ones(1 << 20, device=dev)
dist.all_reduce(x)                              # RCCL's stream is drawn from the pool here
torch.cuda.synchronize()
if mode == "pool":
    sides = [torch.cuda.Stream() for _ in range(32)]          # every member of the low-priority pool once
    cap = torch.cuda.Stream()
else:
    from butd_detr_amd import graph_audit
    sides = [graph_audit.own_stream(dev, role=f"probe{i}") for i in range(32)]
    cap = graph_audit.own_stream(dev, role="probe.capture")
    pool = {torch.cuda.Stream().cuda_stream for _ in range(64)}
    assert not ({s.cuda_stream for s in sides} | {cap.cuda_stream}) & pool, "an own stream aliases a pool stream"
print(f"mode {mode}: {len({s.cuda_stream for s in sides})} distinct side streams", flush=True)
for i, side in enumerate(sides):
    w = dist.all_reduce(x, async_op=True)       # a fresh collective: its work sits in the watchdog's list for <= 100 ms
    w.wait()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            y = x * 2
        time.sleep(0.3)                         # the watchdog polls while `side` is capturing
        torch.cuda.current_stream().wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    print(f"side stream {i:2d} (handle {side.cuda_stream:#x}): capture + replay ok", flush=True)
dist.destroy_process_group()
print("all 32 offsets passed", flush=True)
