#!/bin/bash
# A/B of one environment switch in the default bench, alternating in ONE gpurun call: ab_env.sh VAR valA valB [reps] [steps]
VAR=$1; A=$2; B=$3; REPS=${4:-3}; STEPS=${5:-60}
cd $GRAFT_REPO_ROOT
for i in $(seq $REPS); do
  for v in $A $B; do
    ms=$(env $VAR=$v python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$VAR=$v $ms"
  done
done
