# refresh profiles/: kernel stats of the default bench command, one-step summary, PMC traffic
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
rm -rf /tmp/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/final/bench_under_rocprof.log 2>&1
cp /tmp/prof/bench_kernel_stats.csv gpurun_out/final/r01_hip_bench_kernel_stats.csv
python scratch/trace_summary.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel 7 > gpurun_out/final/r01_hip_one_step_summary.txt
tail -1 gpurun_out/final/bench_under_rocprof.log | cut -c1-300
head -12 gpurun_out/final/r01_hip_one_step_summary.txt | cut -c1-120
bash scratch/pmc.sh > gpurun_out/final/pmc.log 2>&1
cp gpurun_out/pmc/*.json gpurun_out/final/
cat gpurun_out/pmc/FETCH_SIZE.json
timeout 600 python bench.py > gpurun_out/final/r01_bench_default.json 2> gpurun_out/final/bench_default.err
tail -1 gpurun_out/final/r01_bench_default.json | cut -c1-400
