#!/bin/bash
# round 6: A/B of environment settings ("VAR=value VAR2=value", "-" = none) in the default bench command, alternating
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for s in "$@"; do
    e="$s"; [ "$s" = "-" ] && e=""
    r=$(env $e BUTD_BENCH_NO_CHILD=1 timeout 600 python bench.py --steps 60 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['final_loss'])")
    echo "[$s] $r"
  done
done
