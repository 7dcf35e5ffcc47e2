cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$C -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pmc_$C.log 2>&1
  ls /tmp/pmc_$C | head
  python - <<PY
import csv, collections, json
rows = list(csv.DictReader(open("/tmp/pmc_$C/pmc_counter_collection.csv")))
print(rows[0].keys() if rows else "no rows")
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if r.get("Counter_Name") != "$C": continue
    name = r["Kernel_Name"]
    key = "gemm_kernel" if "gemm_kernel" in name else "ball_query_kernel" if "ball_query_kernel" in name else "attn_fwd_kernel" if "attn_fwd_kernel" in name else "fps_pruned_kernel" if "fps_pruned" in name else "bq_grid_query" if "bq_grid_query" in name else "lsap_kernel" if "lsap_kernel" in name else None
    if key:
        agg[key][0] += 1; agg[key][1] += float(r["Counter_Value"])
out = {k: {"launches": c, "avg_$C": v / c} for k, (c, v) in agg.items()}
print(out)
json.dump(out, open("gpurun_out/pmc/$C.json", "w"), indent=1)
PY
done
