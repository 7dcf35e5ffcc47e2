"""-m gpu: the streams of the captured step are not members of torch's stream pool.

Root cause of round 4's unexplained abort (DESIGN.md 7.4 #6b: RCCL's watchdog thread terminating the process with
'operation not permitted on an event last recorded in a capturing stream' during a two-piece capture, in one test order
only): ``torch.cuda.Stream()`` hands out the 32 members of a per-priority pool round-robin, ProcessGroupNCCL's stream is
one of them, and after enough Stream() calls in a process a "new" side stream of the step IS RCCL's stream.  Forked inside
a capture it puts RCCL's stream into capture mode while the watchdog polls the end event of the last eager collective.
scratch/stream_alias_probe.py walks all 32 pool members as the forked stream of a capture next to a live process group:
with pool streams the process aborts at the aliasing member (profiles/r05_stream_alias.txt), with
``graph_audit.own_stream`` it cannot."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_own_streams_are_outside_the_pool_and_stable():
    from butd_detr_amd import graph_audit
    dev = torch.device("cuda", 0)
    mine = [graph_audit.own_stream(dev, role=f"test.alias.{i}") for i in range(4)]
    pool = {torch.cuda.Stream(dev).cuda_stream for _ in range(96)}          # every member of the pool, three times over
    assert len(pool) <= 64                                                   # (it IS a pool: 96 "new" streams, <= 32 handles per priority)
    assert len({s.cuda_stream for s in mine}) == 4
    assert not {s.cuda_stream for s in mine} & pool
    assert graph_audit.own_stream(dev, role="test.alias.0").cuda_stream == mine[0].cuda_stream   # one per role, kept
    x = torch.ones(1 << 16, device=dev)
    mine[0].wait_stream(torch.cuda.current_stream())          # (own streams are non-blocking: order the read after the fill)
    with torch.cuda.stream(mine[0]):
        y = x * 3
    torch.cuda.current_stream().wait_stream(mine[0])
    assert float(y.sum()) == 3 * (1 << 16)


def test_a_capture_forking_every_own_stream_next_to_rccl_survives_the_watchdog():
    """The reproducer with the step's own streams: 32 captures, each forking another stream, a fresh collective in front
    of every capture and 0.3 s inside it for the watchdog to poll.  (``pool`` instead of ``own`` aborts the child.)"""
    import socket
    with socket.socket() as sock:              # a free rendezvous port (as bench._spawn_ranks picks one)
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scratch", "stream_alias_probe.py"), "own"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    assert "all 32 offsets passed" in out.stdout
