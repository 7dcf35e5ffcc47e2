mkdir -p gpurun_out/r2c21
timeout 1200 python -m pytest tests/test_gpu_fused_attention.py -x -q -m gpu 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_gpu_timed_shapes.py -x -q -m gpu -k bf16 2>&1 | tail -3
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2c21/bench.json 2> gpurun_out/r2c21/bench.err; echo "bench rc=$?"; tail -2 gpurun_out/r2c21/bench.err | cut -c1-200
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2c21/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline_attention']['achieved'], d['roofline_attention']['fwd_ms'], d['roofline_attention']['bwd_ms'])
b=d['bf16_operating_point']; print(b['value'], b['ms_per_step'], b['roofline']['achieved'], b['roofline_attention'])
PY
