#!/bin/bash
# round 6: kernel trace of the default bench command (no extras) -> gpurun_out/r06/trace_<tag>/ ; window listing of one step's main queue
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-a}; LO=${2:-9.0}; HI=${3:-11.5}
O=gpurun_out/r06; mkdir -p $O
rm -rf /tmp/prof_$TAG
BUTD_BENCH_NO_CHILD=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras > $O/trace_$TAG.log 2>&1
cp /tmp/prof_$TAG/bench_kernel_stats.csv $O/kernel_stats_$TAG.csv
python scratch/window_kernels.py /tmp/prof_$TAG/bench_kernel_trace.csv fps_pruned_kernel 2 $LO $HI > $O/window_$TAG.txt
python scratch/window_kernels.py /tmp/prof_$TAG/bench_kernel_trace.csv fps_pruned_kernel 2 0 40 > $O/window_full_$TAG.txt
python scratch/small_kernels.py /tmp/prof_$TAG/bench_kernel_trace.csv fps_pruned_kernel 2 > $O/small_kernels_$TAG.txt
python scratch/torch_kernels_on_main.py /tmp/prof_$TAG/bench_kernel_trace.csv fps_pruned_kernel 2 > $O/stock_kernels_$TAG.txt
tail -1 $O/trace_$TAG.log | cut -c1-200
