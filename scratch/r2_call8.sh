mkdir -p gpurun_out/r2c8
timeout 900 python -m pytest tests/test_gpu_gemm_fuzz.py -x -q -m gpu 2>&1 | tail -1
echo "== r1 lib"; BUTD_HIP_LIB=$PWD/scratch/exp/libr1.so TILES="0x0" timeout 900 python scratch/gemm_cases.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2c8/gemm_r1.txt; cat gpurun_out/r2c8/gemm_r1.txt
echo "== r2 lib"; timeout 900 python scratch/gemm_cases.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2c8/gemm_r2.txt; cat gpurun_out/r2c8/gemm_r2.txt
