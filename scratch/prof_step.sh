cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > /tmp/bench.log 2>&1
python scratch/trace_summary.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel ${SKIP:-7} > gpurun_out/step_summary.txt
tail -8 gpurun_out/step_summary.txt
