mkdir -p gpurun_out/r2c4
timeout 900 python -m pytest tests/test_gpu_gemm_fuzz.py -x -q -m gpu 2>&1 | tail -2
export TILES="64x64,128x64,64x96,96x96,128x96,32x32,32x96"
for v in prod p0 p1 p2s1 p2s2 p2a1 p2a4 p2a16; do
  if [ $v = prod ]; then unset BUTD_HIP_LIB; else export BUTD_HIP_LIB=$PWD/scratch/exp/libabl_$v.so; fi
  echo "== variant $v"
  CASES="fwd 3x(8192" timeout 300 python scratch/gemm_cases.py 2>&1 | grep -v "amdgpu.ids\|^case"
  CASES="fwd 3x(2048" timeout 300 python scratch/gemm_cases.py 2>&1 | grep -v "amdgpu.ids\|^case"
  CASES="dgrad+wgrad 8192" timeout 300 python scratch/gemm_cases.py 2>&1 | grep -v "amdgpu.ids\|^case"
  CASES="SA fwd (256k,256,128)" timeout 300 python scratch/gemm_cases.py 2>&1 | grep -v "amdgpu.ids\|^case"
  CASES="SA bwd w(256,128,256k)" timeout 300 python scratch/gemm_cases.py 2>&1 | grep -v "amdgpu.ids\|^case"
done > gpurun_out/r2c4/variants.txt 2>&1
cat gpurun_out/r2c4/variants.txt
unset BUTD_HIP_LIB; unset TILES
timeout 600 python scratch/diag_train6.py > gpurun_out/r2c4/diag_train6.txt 2>&1
cat gpurun_out/r2c4/diag_train6.txt | grep -v Warn
