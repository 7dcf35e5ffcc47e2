cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof; mkdir -p gpurun_out/prof_r01
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r01/bench_stdout.log 2>&1
tail -1 gpurun_out/prof_r01/bench_stdout.log | cut -c1-200
cp /tmp/prof/bench_kernel_stats.csv gpurun_out/prof_r01/r01_hip_bench_kernel_stats.csv
python scratch/trace_summary.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel 7 > gpurun_out/prof_r01/r01_hip_one_step_summary.txt
head -12 gpurun_out/prof_r01/r01_hip_one_step_summary.txt
