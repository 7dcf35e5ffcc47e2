mkdir -p gpurun_out/r2c19
timeout 900 python -m pytest tests/test_gpu_gemm_fuzz.py -x -q -m gpu -k "tall" 2>&1 | tail -6
timeout 900 python scratch/tall_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r2c19/tall.txt; cat gpurun_out/r2c19/tall.txt
