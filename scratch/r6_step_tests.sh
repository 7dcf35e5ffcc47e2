#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
BUTD_BENCH_NO_CHILD=1 timeout 600 python bench.py --steps 60 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-230
timeout 1200 python -m pytest $(grep -ln 'GraphedTrainStep' tests/*.py) -x -q -m gpu > $O/step_tests.log 2>&1; tail -3 $O/step_tests.log
BUTD_BENCH_NO_CHILD=1 timeout 600 python bench.py --steps 60 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-230
