#!/bin/bash
# A/B of fused_mlp._THIN_SPLIT (slices of the thin weight gradients) in the step: ab_thin.sh "8 16 32" reps
cd $GRAFT_REPO_ROOT
for i in $(seq ${2:-2}); do
  for v in $1; do
    ms=$(BUTD_BENCH_NO_CHILD=1 BUTD_THIN=$v python -c "
import os, sys, runpy
from butd_detr_amd import fused_mlp
fused_mlp._THIN_SPLIT[0] = int(os.environ['BUTD_THIN'])
sys.argv = ['bench.py', '--steps', '60', '--warmup', '5', '--no-cpu-baseline', '--no-extras']
runpy.run_path('bench.py', run_name='__main__')
" 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "thin_split=$v $ms"
  done
done
