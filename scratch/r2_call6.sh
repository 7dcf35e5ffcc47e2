mkdir -p gpurun_out/r2c6
timeout 900 python -m pytest tests/test_gpu_gemm_fuzz.py tests/test_gpu_fused_attention.py -x -q -m gpu 2>&1 | tail -2
TILES="0x0,64x64,64x-64,64x96,96x32,128x96" timeout 900 python scratch/gemm_cases.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2c6/gemm_auto.txt
cat gpurun_out/r2c6/gemm_auto.txt
timeout 300 python scratch/attn_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2c6/attn_bench.txt; cat gpurun_out/r2c6/attn_bench.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2c6/bench.json 2> gpurun_out/r2c6/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2c6/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline_attention']['achieved'], d['roofline_attention']['fwd_ms'], d['roofline_attention']['bwd_ms'])
PY
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_gemm_fuzz.py --deselect tests/test_gpu_fused_attention.py > gpurun_out/r2c6/gpu_tests.log 2>&1; echo "tests rc=$?"
tail -8 gpurun_out/r2c6/gpu_tests.log
