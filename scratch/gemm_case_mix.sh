# dynamic instruction mix of butd_gemm_grouped per case of scratch/gemm_cases.py (SQ counters, one pass);
# LIBS="tag:path ..." compares ablation builds (scratch/build_abl.sh gemm_ops ...)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for spec in ${LIBS:-product:butd_detr_amd/lib/libbutd_detr_hip.so}; do
  tag=${spec%%:*}; lib=${spec#*:}
  rm -rf /tmp/cmix_$tag
  BUTD_HIP_LIB=$lib CASE_META=/tmp/case_meta.json timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d /tmp/cmix_$tag -o c -- python scratch/gemm_case_pmc.py > /tmp/cmix_$tag.log 2>&1
done
python - <<'PY'
import csv, json, collections, os
meta = json.load(open("/tmp/case_meta.json"))
specs = os.environ.get("LIBS", "product:x").split()
cols = {}
for spec in specs:
    tag = spec.split(":")[0]
    rows = [r for r in csv.DictReader(open(f"/tmp/cmix_{tag}/c_counter_collection.csv")) if "gemm_kernel" in r["Kernel_Name"]]
    by = collections.OrderedDict()
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        by.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
    d = list(by.values())
    assert len(d) == 3 * len(meta), (tag, len(d), len(meta))
    cols[tag] = [d[3 * i + 2] for i in range(len(meta))]
first = specs[0].split(":")[0]
print("non-MFMA VALU instructions per ideal MFMA (flops / 2048), per build; SALU and LDS of the first build")
print("%-40s %-22s " % ("case", "kernel") + " ".join("%9s" % s.split(":")[0] for s in specs) + " %9s %9s" % ("SALU", "LDS"))
for i, m in enumerate(meta):
    x = cols[first][i]; n = x["name"]; t = n[n.index("gemm_kernel<") + 12:n.index(">")]
    ideal = m["flops"] / 2048
    vals = [(cols[s.split(":")[0]][i]["SQ_INSTS_VALU"] - cols[s.split(":")[0]][i]["SQ_INSTS_MFMA"]) / ideal for s in specs]
    print("%-40s %-22s " % (m["case"], t) + " ".join("%9.2f" % v for v in vals) + " %9.2f %9.2f" % (x["SQ_INSTS_SALU"] / ideal, x["SQ_INSTS_LDS"] / ideal))
PY
