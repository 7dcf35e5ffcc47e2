mkdir -p gpurun_out/r2c7
echo "== r1 lib"; BUTD_HIP_LIB=$PWD/scratch/exp/libr1.so TILES="0x0" timeout 900 python scratch/gemm_cases.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2c7/gemm_r1.txt; cat gpurun_out/r2c7/gemm_r1.txt
echo "== r2 lib"; TILES="0x0" timeout 900 python scratch/gemm_cases.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2c7/gemm_r2.txt; cat gpurun_out/r2c7/gemm_r2.txt
echo "== r1 attn"; BUTD_HIP_LIB=$PWD/scratch/exp/libr1.so timeout 300 python scratch/attn_bench.py 2>&1 | grep "p=0.1"
echo "== r1 lib bench"; BUTD_HIP_LIB=$PWD/scratch/exp/libr1.so timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_gemm_fuzz.py --deselect tests/test_gpu_fused_attention.py > gpurun_out/r2c7/gpu_tests.log 2>&1; echo "tests rc=$?"
grep -E "^FAILED|passed|failed" gpurun_out/r2c7/gpu_tests.log | head -30
