#!/bin/bash
# kernel stats + one-step summary of the default bench command (no extras): gpurun_out/quick/<tag>_*
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-q}; O=gpurun_out/quick; mkdir -p $O
export BUTD_BENCH_NO_CHILD=1
rm -rf /tmp/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/${TAG}_bench_under_rocprof.log 2>&1
cp /tmp/prof/bench_kernel_stats.csv $O/${TAG}_kernel_stats.csv
python scratch/trace_summary.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel ${SKIP:-2} > $O/${TAG}_one_step_summary.txt
python scratch/step_timeline.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel ${SKIP:-2} 10 > $O/${TAG}_step_timeline.txt
tail -1 $O/${TAG}_bench_under_rocprof.log | cut -c1-200
head -40 $O/${TAG}_one_step_summary.txt | cut -c1-150; tail -7 $O/${TAG}_one_step_summary.txt
