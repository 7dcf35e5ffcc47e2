#!/bin/bash
# A/B timing of step options without the profiler: ms_per_step of the default bench
cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-bf16-row 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
run A=default
run BUTD_ENCODER_FORK=0
run A=default
run BUTD_ENCODER_FORK=0
