for t in base nomfma noload d8 d2; do
  if [ $t = base ]; then unset BUTD_HIP_LIB; else export BUTD_HIP_LIB=$GRAFT_REPO_ROOT/scratch/exp/libabl_$t.so; fi
  echo "== $t"; python scratch/panel_bench.py 2>&1 | grep "rows 2048 R=16\|rows 8192 R=32"
done
