set -x
mkdir -p gpurun_out/r2c1
timeout 900 python -m pytest tests/test_gpu_gemm_fuzz.py -x -q -m gpu > gpurun_out/r2c1/fuzz.log 2>&1; echo "fuzz rc=$?" 
tail -5 gpurun_out/r2c1/fuzz.log
timeout 1200 python scratch/gemm_cases.py > gpurun_out/r2c1/gemm_cases.txt 2>&1; echo "cases rc=$?"
cat gpurun_out/r2c1/gemm_cases.txt
timeout 1200 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_gemm_fuzz.py > gpurun_out/r2c1/gpu_tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r2c1/gpu_tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2c1/bench.json 2> gpurun_out/r2c1/bench.err; echo "bench rc=$?"
cat gpurun_out/r2c1/bench.json | cut -c1-1500
