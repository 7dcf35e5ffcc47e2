mkdir -p gpurun_out/r2c20
for fork in 1 0; do
  echo "== BUTD_ENCODER_FORK=$fork"
  BUTD_ENCODER_FORK=$fork timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-bf16-row 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['final_loss'])"
done
timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_gpu_gemm_fuzz.py -x > gpurun_out/r2c20/gpu_tests.log 2>&1; echo "tests rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r2c20/gpu_tests.log | head -20
