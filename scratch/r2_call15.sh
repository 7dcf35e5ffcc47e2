mkdir -p gpurun_out/r2c15
timeout 1200 python -m pytest tests/test_gpu_gemm_fuzz.py tests/test_aten_binding.py -x -q -m gpu 2>&1 | tail -4
timeout 1200 python -m pytest tests/test_gpu_timed_shapes.py -x -q -m gpu -k bf16 2>&1 | tail -8
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2c15/bench.json 2> gpurun_out/r2c15/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r2c15/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2c15/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_attention']['achieved'])
print(d.get('bf16_operating_point'))
PY
