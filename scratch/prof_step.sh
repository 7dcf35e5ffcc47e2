cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > /tmp/bench.log 2>&1
tail -1 /tmp/bench.log | cut -c1-200
grep -c . /tmp/prof/bench_kernel_trace.csv
grep "lsap" /tmp/prof/bench_kernel_stats.csv | cut -c1-200
python scratch/trace_summary.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel ${SKIP:-7} > gpurun_out/step_summary.txt
head -3 gpurun_out/step_summary.txt | cut -c1-170
grep -i "elementwise\|reduce\|lsap\|softmax\|scatter\|gather\|index\|topk\|sort\|where\|bmm\|Cijk" gpurun_out/step_summary.txt | cut -c1-200 | head -40
