#!/bin/bash
# round 6, second session: state at HEAD -- GPU suite, default bench line, kernel stats of the bf16 operating point
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputests_a.log 2>&1; tail -3 $O/gputests_a.log
timeout 600 python bench.py > $O/bench_default_a.json 2> $O/bench_default_a.err; tail -1 $O/bench_default_a.json | cut -c1-400
rm -rf /tmp/prof_bf
BUTD_BENCH_NO_CHILD=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bf -o bench -- python bench.py --dtype bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-extras > $O/trace_bf16.log 2>&1
cp /tmp/prof_bf/bench_kernel_stats.csv $O/kernel_stats_bf16.csv
python scratch/trace_summary.py /tmp/prof_bf/bench_kernel_trace.csv fps_pruned_kernel 2 > $O/one_step_summary_bf16.txt
python scratch/step_timeline.py /tmp/prof_bf/bench_kernel_trace.csv fps_pruned_kernel 2 10 > $O/step_timeline_bf16.txt
tail -1 $O/trace_bf16.log | cut -c1-200
head -40 $O/one_step_summary_bf16.txt | cut -c1-150
