#!/bin/bash
# round 6: A/B of BUTD_AB settings in the default bench command (no extras): ms_per_step of each setting, alternating
# usage: scratch/r6_ab.sh "setting A" "setting B" ... (a setting is a BUTD_AB string; "-" = product defaults)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for s in "$@"; do
    ab="$s"; [ "$s" = "-" ] && ab=""
    r=$(BUTD_AB="$ab" BUTD_BENCH_NO_CHILD=1 timeout 600 python bench.py --steps 60 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['final_loss'])")
    echo "[$s] $r"
  done
done
