#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_golden_modules.py -x -q -m gpu > $O/golden_tests_$i.log 2>&1; tail -2 $O/golden_tests_$i.log; cp gpurun_out/golden_errors.json $O/golden_errors_$i.json; done
