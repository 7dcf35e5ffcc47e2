#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fps_prefix.py tests/test_gpu_pointnet2_parity.py tests/test_gpu_stress_config4.py tests/test_gpu_function_api.py tests/test_aten_binding.py -x -q -s -m gpu > $O/fps_tests.log 2>&1; grep -n 'FPS levels\|passed\|failed\|Error\|error' $O/fps_tests.log | head -20
BUTD_BENCH_NO_CHILD=1 timeout 600 python bench.py --steps 60 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-220
