cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > /tmp/bench.log 2>&1
python scratch/trace_summary.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel 7 > gpurun_out/step_summary.txt
grep -i "gather_segments\|multi_tensor\|step wall" gpurun_out/step_summary.txt | cut -c1-200
