cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-row > /tmp/b.log 2>&1
python scratch/trace_summary.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel 7 > gpurun_out/step_summary.txt
head -42 gpurun_out/step_summary.txt | cut -c1-150
