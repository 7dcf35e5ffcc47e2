# run-to-run reproducibility of the graphed training trajectory per configuration (loss at step 60, twice each)
for cfg in "BUTD_FAN_OUT=0 BUTD_PROJ_CHAIN=0" "BUTD_FAN_OUT=1 BUTD_PROJ_CHAIN=0" "BUTD_FAN_OUT=0 BUTD_PROJ_CHAIN=1" "BUTD_FAN_OUT=1 BUTD_PROJ_CHAIN=1" "BUTD_FAN_OUT=0 BUTD_PROJ_CHAIN=0 BUTD_ENCODER_FORK=0" "BUTD_FAN_OUT=0 BUTD_PROJ_CHAIN=0 PT=0" "BUTD_FAN_OUT=0 BUTD_PROJ_CHAIN=0 PS=0"; do
  for r in 1 2; do
    echo -n "$cfg run $r: "
    env $cfg STEPS=60 python scratch/soak_cfg.py 2>&1 | grep "prefetch" | grep -o '\[[0-9., ]*\]$'
  done
done
