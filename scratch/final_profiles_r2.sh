# round-2 profiles: kernel stats of the default bench command, one-step summary, PMC traffic (whole step and per
# GEMM case), per-shape GEMM tables, attention counters.  Everything lands in gpurun_out/final_r2/ ; the summaries
# are copied into profiles/ afterwards.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r2; mkdir -p $O
rm -rf /tmp/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-row > $O/bench_under_rocprof.log 2>&1
cp /tmp/prof/bench_kernel_stats.csv $O/r02_hip_bench_kernel_stats.csv
python scratch/trace_summary.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel 7 > $O/r02_hip_one_step_summary.txt
tail -1 $O/bench_under_rocprof.log | cut -c1-200
head -14 $O/r02_hip_one_step_summary.txt | cut -c1-130; tail -7 $O/r02_hip_one_step_summary.txt
# PMC: whole step, separate passes
mkdir -p gpurun_out/pmc
bash scratch/pmc.sh > $O/pmc.log 2>&1
cp gpurun_out/pmc/FETCH_SIZE.json gpurun_out/pmc/WRITE_SIZE.json $O/
# PMC per GEMM case
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cpmc_$C
  CASE_META=/tmp/case_meta.json timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/cpmc_$C -o c -- python scratch/gemm_case_pmc.py > /tmp/cpmc_$C.log 2>&1
done
python - <<'PY' > gpurun_out/final_r2/r02_gemm_case_traffic.txt
import csv, json
meta = json.load(open("/tmp/case_meta.json"))
def series(C):
    rows = [r for r in csv.DictReader(open(f"/tmp/cpmc_{C}/c_counter_collection.csv")) if r["Counter_Name"] == C and "gemm_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [(float(r["Counter_Value"]), r["Kernel_Name"]) for r in rows]
f, w = series("FETCH_SIZE"), series("WRITE_SIZE")
print("per-case HBM traffic of butd_gemm_grouped (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024,")
print("MI355X_MICROARCH.md HBM section); last of three launches of the case; algorithmic = every operand and the result once")
print("%-40s %-26s %12s %12s %7s" % ("case", "kernel <TM,TN,FAST,PIPE>", "measured MB", "algorithmic", "ratio"))
assert len(f) == len(w) == 3 * len(meta), (len(f), len(w), len(meta))
for i, m in enumerate(meta):
    fk, name = f[3 * i + 2]; wk, _ = w[3 * i + 2]
    mb = (2 * fk + wk) * 1024 / 1e6
    tmpl = name[name.index("gemm_kernel<") + 12:name.index(">")]
    print("%-40s %-26s %12.1f %12.1f %7.2f" % (m["case"], tmpl, mb, m["algorithmic_bytes"] / 1e6, mb / (m["algorithmic_bytes"] / 1e6)))
PY
cat $O/r02_gemm_case_traffic.txt | cut -c1-140
# per-shape tables
timeout 600 python scratch/gemm_shapes.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > $O/r02_gemm_shapes.txt
timeout 900 python scratch/gemm_cases.py 2>&1 | grep -v amdgpu.ids > $O/r02_gemm_tiles.txt
head -3 $O/r02_gemm_shapes.txt
# attention counters
bash scratch/attn_pmc.sh > $O/r02_attn_pmc.txt 2>&1; tail -30 $O/r02_attn_pmc.txt
timeout 900 python bench.py > $O/r02_bench_default.json 2> $O/bench_default.err
tail -1 $O/r02_bench_default.json | cut -c1-300
