cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 3 --warmup 2 --backend hip --no-cpu-baseline > /tmp/bench_prof.log 2>&1
 python scratch/trace_head.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel ${1:-0} ${2:-1}
