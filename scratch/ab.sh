#!/bin/bash
# A/B timing of step options without the profiler: ms_per_step of the default bench
cd $GRAFT_REPO_ROOT
export BUTD_BENCH_NO_CHILD=1
run() { echo -n "$* : "; env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['launches_per_step'])"; }
for v in "$@"; do run $v; run A=default; done
