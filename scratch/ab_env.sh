#!/bin/bash
# A/B of the default bench under two environments, alternating in ONE gpurun call (no profiler, no roofline child):
#   ab_env.sh "BUTD_AB=wgrad_slabs=0" "BUTD_AB=" [reps] [steps]          (any VAR=value works; BUTD_AB: butd_detr_amd/switches.py)
A=$1; B=$2; REPS=${3:-3}; STEPS=${4:-60}
cd $GRAFT_REPO_ROOT
for i in $(seq $REPS); do
  for v in "$A" "$B"; do
    ms=$(env "$v" BUTD_BENCH_NO_CHILD=1 python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$v $ms"
  done
done
