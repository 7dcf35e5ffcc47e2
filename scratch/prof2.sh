cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/* /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 3 --warmup 2 --backend ${1:-torch} --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1
tail -1 gpurun_out/bench_prof.log | cut -c1-200
python scratch/trace_summary.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel 7 | head -${2:-70}
