mkdir -p gpurun_out/r2c18
timeout 900 python -m pytest tests/test_gpu_gemm_fuzz.py -x -q -m gpu 2>&1 | tail -1
TILES="0x0,32x-32,32x32,32x1032,64x64,64x1064,64x-64" timeout 900 python scratch/gemm_cases.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2c18/pipe4.txt; cat gpurun_out/r2c18/pipe4.txt
