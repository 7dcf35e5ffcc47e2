mkdir -p gpurun_out/r2c12
timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_gemm_fuzz.py > gpurun_out/r2c12/gpu_tests.log 2>&1; echo "tests rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r2c12/gpu_tests.log | head -30
tail -40 gpurun_out/r2c12/gpu_tests.log | cut -c1-220
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r2c12/bench.json 2> gpurun_out/r2c12/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r2c12/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2c12/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_attention']['achieved'])
print(json.dumps(d['roofline_ball_query'])[:900]); print(d['matcher_detection_split'], d['cpu_baseline'])
PY
