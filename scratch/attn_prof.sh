cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/ap
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ap -o a -- python scratch/attn_bench.py > /tmp/ap.log 2>&1
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("/tmp/ap/a_kernel_trace.csv")))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "attn_" in n:
        import re
        key = re.search(r"attn_\w+<[^>]*>", n).group(0) + f" grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} wg {r['Workgroup_Size_X']}"
        d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = sorted(v); print(f"{k:75s} n={len(v):4d} median {v[len(v)//2]:7.1f} us")
PY
