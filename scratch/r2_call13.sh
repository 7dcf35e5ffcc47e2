mkdir -p gpurun_out/r2c13
timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_gpu_gemm_fuzz.py > gpurun_out/r2c13/gpu_tests.log 2>&1; echo "tests rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r2c13/gpu_tests.log | head -30
