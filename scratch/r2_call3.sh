mkdir -p gpurun_out/r2c3
export CASES="fwd 3x(8192"
export TILES="64x64,128x64,64x96,96x96,128x96,32x32"
for m in 0 1 4 16 30 15; do
  if [ $m = 0 ]; then unset BUTD_HIP_LIB; else export BUTD_HIP_LIB=$PWD/scratch/exp/libabl_$m.so; fi
  echo "== ablation mask $m"
  timeout 300 python scratch/gemm_cases.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r2c3/abl.txt 2>&1
cat gpurun_out/r2c3/abl.txt
unset BUTD_HIP_LIB; unset CASES; unset TILES
timeout 900 python -m pytest tests/test_gpu_gemm_fuzz.py -x -q -m gpu 2>&1 | tail -2
timeout 1200 python scratch/gemm_cases.py > gpurun_out/r2c3/gemm_cases.txt 2>&1
cat gpurun_out/r2c3/gemm_cases.txt
timeout 1500 python -m pytest tests/test_gpu_timed_shapes.py tests/test_gpu_golden_modules.py -x -q -m gpu > gpurun_out/r2c3/newtests.log 2>&1; echo "newtests rc=$?"
tail -15 gpurun_out/r2c3/newtests.log
