cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-row > /tmp/b.log 2>&1
python scratch/trace_window.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel gpurun_out/trace_window.csv
