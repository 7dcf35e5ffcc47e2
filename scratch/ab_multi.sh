#!/bin/bash
# the default bench under N environments, alternating in ONE gpurun call: ab_multi.sh REPS STEPS "ENV1" "ENV2" ...
# (ENV = one VAR=value; "X=" for the baseline)
REPS=$1; STEPS=$2; shift 2
cd $GRAFT_REPO_ROOT
for i in $(seq $REPS); do
  for v in "$@"; do
    ms=$(env "$v" BUTD_BENCH_NO_CHILD=1 python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$v $ms"
  done
done
