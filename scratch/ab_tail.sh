#!/bin/bash
# A/B of losses._TAIL (no BUTD_AB name: a test-only switch) through an env var read by this script's python shim
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for v in 0 1; do
    ms=$(BUTD_BENCH_NO_CHILD=1 BUTD_TAIL=$v python -c "
import os, sys, runpy
from butd_detr_amd import losses
losses._TAIL[0] = os.environ['BUTD_TAIL'] == '1'
sys.argv = ['bench.py', '--steps', '60', '--warmup', '5', '--no-cpu-baseline', '--no-extras']
runpy.run_path('bench.py', run_name='__main__')
" 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "tail=$v $ms"
  done
done
