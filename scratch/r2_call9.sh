mkdir -p gpurun_out/r2c9
timeout 900 python -m pytest tests/test_gpu_gemm_fuzz.py -x -q -m gpu 2>&1 | tail -1
timeout 1500 python scratch/gemm_cases.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2c9/gemm_r2.txt; cat gpurun_out/r2c9/gemm_r2.txt
