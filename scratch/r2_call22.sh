mkdir -p gpurun_out/r2c22
timeout 900 python -m pytest tests/test_gpu_fused_sa.py -x -q -m gpu 2>&1 | tail -6
timeout 300 python scratch/sa_eval_traffic.py 2>&1 | grep -v amdgpu
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/sapmc_$C
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/sapmc_$C -o c -- python scratch/sa_eval_traffic.py pmc > /tmp/sapmc_$C.log 2>&1
done
python - <<'PY' | tee gpurun_out/r2c22/sa_eval_traffic.txt
import csv
def rows(C):
    r = [x for x in csv.DictReader(open(f"/tmp/sapmc_{C}/c_counter_collection.csv")) if x["Counter_Name"] == C]
    r.sort(key=lambda x: int(x["Dispatch_Id"]))
    return r
f, w = rows("FETCH_SIZE"), rows("WRITE_SIZE")
# the script runs one(), many() once for the comparison print, then one(), many() again: take the LAST occurrence
def last_block(rs):
    names = [x["Kernel_Name"] for x in rs]
    i = max(k for k, n in enumerate(names) if "sa_fused_eval_kernel" in n)
    return i
i = last_block(f)
print("SA1 forward, eval, 8 x 50 000 points (2048 centres x 64 neighbours): HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024")
tot = lambda lo, hi: sum((2 * float(f[k]["Counter_Value"]) + float(w[k]["Counter_Value"])) * 1024 for k in range(lo, hi))
print("  one kernel (sa_fused_eval_kernel)      : %8.1f MB" % (tot(i, i + 1) / 1e6))
print("  multi-launch pipeline (%2d kernels after) : %8.1f MB" % (len(f) - i - 1, tot(i + 1, len(f)) / 1e6))
for k in range(i + 1, len(f)):
    print("      %-60s %8.1f MB" % (f[k]["Kernel_Name"][:60], tot(k, k + 1) / 1e6))
PY
