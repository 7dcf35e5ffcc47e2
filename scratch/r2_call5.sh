mkdir -p gpurun_out/r2c5
export TILES="32x32,64x64,32x96,64x96,96x32,128x64,128x96"
for v in prod p0; do
  if [ $v = prod ]; then unset BUTD_HIP_LIB; else export BUTD_HIP_LIB=$PWD/scratch/exp/libabl_$v.so; fi
  echo "== variant $v"
  timeout 900 python scratch/gemm_cases.py 2>&1 | grep -v "amdgpu.ids"
done > gpurun_out/r2c5/variants.txt 2>&1
cat gpurun_out/r2c5/variants.txt
unset BUTD_HIP_LIB; unset TILES
timeout 600 python scratch/diag_train6.py 2>&1 | grep -v Warn | grep "==\|center\|grad" > gpurun_out/r2c5/diag_train6.txt
cat gpurun_out/r2c5/diag_train6.txt
timeout 1500 python -m pytest tests/test_gpu_timed_shapes.py -x -q -m gpu > gpurun_out/r2c5/timed.log 2>&1; echo "timed rc=$?"
tail -15 gpurun_out/r2c5/timed.log
