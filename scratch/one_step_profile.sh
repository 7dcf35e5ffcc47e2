#!/bin/bash
# kernel trace of the default bench (no extras) -> one-step summary + timeline in gpurun_out/r05/ (suffix $1)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export BUTD_BENCH_NO_CHILD=1
rm -rf /tmp/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_under_rocprof$1.log 2>&1
python scratch/trace_summary.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel ${SKIP:-2} > $O/one_step_summary$1.txt
python scratch/step_timeline.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel ${SKIP:-2} 10 > $O/step_timeline$1.txt
python scratch/torch_kernels_on_main.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel ${SKIP:-2} > $O/stock_kernels$1.txt
python scratch/small_kernels.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel ${SKIP:-2} > $O/small_kernels$1.txt
if [ -n "$WIN" ]; then python scratch/window_kernels.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel ${SKIP:-2} $WIN > $O/window$1.txt; fi
tail -1 $O/bench_under_rocprof$1.log | cut -c1-200
head -${HEAD:-48} $O/one_step_summary$1.txt | cut -c1-150; tail -7 $O/one_step_summary$1.txt; cat $O/stock_kernels$1.txt | cut -c1-330
