#!/bin/bash
# round-3 profile of the default bench command: kernel stats, one-step summary, timeline, residue
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r3; mkdir -p $O
rm -rf /tmp/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.log 2>&1
cp /tmp/prof/bench_kernel_stats.csv $O/hip_bench_kernel_stats.csv
python scratch/trace_summary.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel ${SKIP:-7} > $O/hip_one_step_summary.txt
python scratch/step_timeline.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel ${SKIP:-7} 10 > $O/step_timeline.txt
python scratch/trace_residue.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel > $O/trace_residue.txt
tail -1 $O/bench_under_rocprof.log | cut -c1-300
