#!/bin/bash
cd /root/repo
export BUTD_STEP_VERBOSE=1
run() { echo "== $*"; env "$@" timeout 300 python scratch/race_probe3.py > /tmp/o.txt 2>&1; grep -E "LR0|loss seq|GraphedTrainStep|fault" /tmp/o.txt | cut -c1-600; grep -qE "LR0|loss seq" /tmp/o.txt || tail -8 /tmp/o.txt; }
run TAG=S1_inline LR0=1 PS=0 PT=0
run TAG=S1b_inline_split LR0=1 PS=0 PT=0 SPLIT=1 OVERLAP=1 BUTD_STEP_SYNC=0
run TAG=S1c_inline_notextov_nofork_split LR0=1 PS=0 PT=0 SPLIT=1 OVERLAP=1 BUTD_STEP_SYNC=0 BUTD_ENCODER_FORK=0 BUTD_TEXT_OVERLAP=0
unset BUTD_STEP_VERBOSE
timeout 1800 python -m pytest tests/test_gpu_free_running.py -x -q -s > /tmp/t.log 2>&1; grep -v "^  File\|^Extension" /tmp/t.log | tail -40
