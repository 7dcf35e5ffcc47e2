cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
for d in scratch/head .; do
  cd $ROOT/$d
  rm -rf /tmp/prof
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 3 --warmup 2 --backend hip --no-cpu-baseline > /tmp/bench_prof.log 2>&1
  echo "=== $d"; tail -1 /tmp/bench_prof.log | cut -c130-170
  python $ROOT/scratch/trace_summary.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel 7 | head -${1:-32} | cut -c1-140
done
