mkdir -p gpurun_out/r2c10
TILES="0x0" timeout 900 python scratch/gemm_cases.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2c10/gemm_auto.txt; cat gpurun_out/r2c10/gemm_auto.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2c10/bench.json 2> gpurun_out/r2c10/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2c10/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline_attention']['achieved'], d['roofline_attention']['fwd_ms'], d['roofline_attention']['bwd_ms'])
PY
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_gemm_fuzz.py > gpurun_out/r2c10/gpu_tests.log 2>&1; echo "tests rc=$?"
grep -E "^FAILED|passed|failed" gpurun_out/r2c10/gpu_tests.log | head -30
