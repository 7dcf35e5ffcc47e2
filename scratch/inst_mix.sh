# dynamic instruction mix per kernel family of one bench run (SQ counters; one pass)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/imix
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d /tmp/imix -o m -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-row > /tmp/imix.log 2>&1
tail -1 /tmp/imix.log | cut -c1-100
python - <<'PY'
import csv, collections, re
rows = list(csv.DictReader(open("/tmp/imix/m_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in rows:
    n = r["Kernel_Name"]
    m = re.search(r"(gemm_kernel<[^>]*>|attn_\w+<[^>]*>|ln_\w+kernel|sa_\w+kernel|mlp_\w+kernel)", n)
    if not m: continue
    k = m.group(1)
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); cnt[k] += 1
print("%-42s %6s %12s %12s %10s %10s %8s" % ("kernel", "n", "VALU", "MFMA", "SALU", "LDS", "VALU/MFMA"))
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_INSTS_MFMA", 0)):
    v, m = d.get("SQ_INSTS_VALU", 0), d.get("SQ_INSTS_MFMA", 0)
    print("%-42s %6d %12.0f %12.0f %10.0f %10.0f %8.2f" % (k, cnt[k], v, m, d.get("SQ_INSTS_SALU", 0), d.get("SQ_INSTS_LDS", 0), (v - m) / m if m else 0))
PY
