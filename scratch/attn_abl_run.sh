for t in base nored noatom noloop noloopnored; do
  if [ $t = base ]; then unset BUTD_HIP_LIB; else export BUTD_HIP_LIB=$GRAFT_REPO_ROOT/scratch/exp/libabl_$t.so; fi
  echo "== $t"; python scratch/attn_short_bench.py 2>&1 | grep "Lq=256\|Lq=1024 Lk=80"
done
