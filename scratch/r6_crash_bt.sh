#!/bin/bash
# native backtrace of the order-dependent crash (graph_step -> free_running -> text_stream in one process)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGUSR1 nostop noprint pass" -ex "handle SIG34 nostop noprint pass" -ex run -ex "bt 40" -ex "info threads" --args python -m pytest tests/test_gpu_graph_step.py tests/test_gpu_free_running.py tests/test_gpu_text_stream.py -x -q -m gpu > gpurun_out/r06/crash_bt.log 2>&1
grep -n "SIGSEGV\|^#" gpurun_out/r06/crash_bt.log | head -60
