#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gputests_final.log 2>&1; tail -3 $O/gputests_final.log
cp gpurun_out/golden_errors.json $O/golden_errors_final.json 2>/dev/null
bash scratch/final_profiles_r6.sh
