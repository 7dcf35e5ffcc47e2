timeout 1200 python -m pytest tests/test_gpu_gemm_fuzz.py -q -m gpu 2>&1 | grep -E "AssertionError|Error|passed|failed" | head
timeout 1200 python -m pytest tests/test_gpu_timed_shapes.py -x -q -m gpu -k bf16 2>&1 | grep -E "AssertionError|^E |passed|failed" | head
