#!/bin/bash
# round 3: bisect of the free-running divergence over the debug hooks
cd /root/repo
export STOCK_DROPOUT=0 STEPS=60
run() { echo "== $*"; env "$@" timeout 300 python scratch/soak_cfg.py 2>&1 | tail -1; }
run TAG=A_default
run TAG=B_fo0 BUTD_FAN_OUT=0
run TAG=C_fo0_sync BUTD_FAN_OUT=0 BUTD_STEP_SYNC=1
run TAG=D_fo0_nofork BUTD_FAN_OUT=0 BUTD_ENCODER_FORK=0
run TAG=E_split_free SPLIT=1 OVERLAP=1 BUTD_STEP_SYNC=0
run TAG=F_split_free_nofork SPLIT=1 OVERLAP=1 BUTD_STEP_SYNC=0 BUTD_ENCODER_FORK=0
run TAG=G_split_nopre_free SPLIT=1 OVERLAP=1 BUTD_STEP_SYNC=0 PS=0 PT=0
run TAG=H_split_nopre_free_nofork SPLIT=1 OVERLAP=1 BUTD_STEP_SYNC=0 PS=0 PT=0 BUTD_ENCODER_FORK=0 BUTD_TEXT_OVERLAP=0
run TAG=I_split_nopre_sync SPLIT=1 OVERLAP=1 BUTD_STEP_SYNC=1 PS=0 PT=0
run TAG=J_split_nopre_sync_nofork SPLIT=1 OVERLAP=1 BUTD_STEP_SYNC=1 PS=0 PT=0 BUTD_ENCODER_FORK=0 BUTD_TEXT_OVERLAP=0
