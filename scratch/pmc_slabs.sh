#!/bin/bash
# HBM traffic of gemm_kernel with the deterministic split-K slabs ON (BUTD_AB=wgrad_slabs=1): the same two PMC passes as
# final_profiles_r5.sh -> gpurun_out/r05/pmc_slabs.json (total bytes of all gemm_kernel launches of the run; the launch count differs
# from the atomic build's -- the flush launches -- so compare TOTALS per step, not per-launch averages)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export BUTD_BENCH_NO_CHILD=1
mkdir -p gpurun_out/r05
for MODE in 0 1; do
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcs_$C
  BUTD_AB=wgrad_slabs=$MODE timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmcs_$C -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /tmp/pmcs_$C.log 2>&1
  python - <<PY
import csv, json
rows = list(csv.DictReader(open("/tmp/pmcs_$C/pmc_counter_collection.csv")))
n, tot = 0, 0.0
for r in rows:
    if r.get("Counter_Name") == "$C" and "gemm_kernel" in r["Kernel_Name"]:
        n += 1; tot += float(r["Counter_Value"])
json.dump({"launches": n, "total_kb": tot}, open("/tmp/pmcs_${MODE}_$C.json", "w"))
PY
done
done
python - <<'PY'
import json
out = {}
for mode in (0, 1):
    f = json.load(open(f"/tmp/pmcs_{mode}_FETCH_SIZE.json")); w = json.load(open(f"/tmp/pmcs_{mode}_WRITE_SIZE.json"))
    out["slabs" if mode else "atomics"] = {"gemm_kernel_launches_in_run": f["launches"],
                                            "hbm_bytes_all_launches": int((2 * f["total_kb"] + w["total_kb"]) * 1024)}
out["ratio_slabs_over_atomics"] = out["slabs"]["hbm_bytes_all_launches"] / out["atomics"]["hbm_bytes_all_launches"]
json.dump(out, open("gpurun_out/r05/pmc_slabs.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
