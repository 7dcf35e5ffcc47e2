#!/bin/bash
# round-6 profiles (everything lands in gpurun_out/final_r6/; the summaries are copied into profiles/ afterwards):
# kernel stats / one-step summary / timeline of the default bench command without its extra sections, the two PMC
# passes, the attention core alone, the default bench line.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r6; mkdir -p $O
export BUTD_BENCH_NO_CHILD=1
rm -rf /tmp/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.log 2>&1
cp /tmp/prof/bench_kernel_stats.csv $O/r06_hip_bench_kernel_stats.csv
python scratch/trace_summary.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel ${SKIP:-2} > $O/r06_hip_one_step_summary.txt
python scratch/step_timeline.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel ${SKIP:-2} 10 > $O/r06_step_timeline.txt
python scratch/torch_kernels_on_main.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel ${SKIP:-2} > $O/r06_stock_kernels.txt
python scratch/small_kernels.py /tmp/prof/bench_kernel_trace.csv fps_pruned_kernel ${SKIP:-2} > $O/r06_small_kernels.txt
tail -1 $O/bench_under_rocprof.log | cut -c1-200
head -5 $O/r06_hip_one_step_summary.txt | cut -c1-130; tail -7 $O/r06_hip_one_step_summary.txt
mkdir -p gpurun_out/pmc
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$C -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /tmp/pmc_$C.log 2>&1
  python - <<PY
import csv, collections, json
rows = list(csv.DictReader(open("/tmp/pmc_$C/pmc_counter_collection.csv")))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if r.get("Counter_Name") != "$C": continue
    name = r["Kernel_Name"]
    key = next((k for k in ("gemm_kernel", "ball_query_kernel", "attn_fwd_kernel", "attn_bwd_dq_kernel", "attn_bwd_dkv_kernel", "attn_bwd_longk_kernel", "attn_dq_fold_kernel", "sa_mid_wide_kernel", "fps_pruned", "bq_grid_query", "lsap_kernel", "ln_bwd_kernel", "sa_colstats", "sa_mask_stats", "sa_dz_mid", "sa_dz_last", "sa_last_fwd", "sa_last_fused", "sa_last_mfma", "sa_last_sparse", "sa_first_stats", "sa_gather_rows", "sa_first_linear_fwd", "sa_first_linear_bwd", "sa_last_coeffs", "fps_prefix_check", "fps_prefix_threshold", "gather_segments") if k in name), None)
    if key:
        agg[key][0] += 1; agg[key][1] += float(r["Counter_Value"])
out = {k: {"launches": c, "avg_$C": v / c} for k, (c, v) in agg.items()}
json.dump(out, open("gpurun_out/pmc/$C.json", "w"), indent=1)
PY
done
python scratch/merge_pmc.py gpurun_out/pmc $O/r06_pmc.json > /dev/null
unset BUTD_BENCH_NO_CHILD
{ echo "# attention core alone (scratch/attn_occ.py, scratch/attn_bench.py): graph-replay timing, B x 8 heads x 1024 x 1024, head dim 36, dropout 0.1"
  python scratch/attn_occ.py 2>/dev/null | grep "B="
  echo "# dropout off / on (what the mask hash costs) and the decoder's shapes:"
  python scratch/attn_bench.py 2>/dev/null | grep "Lq="
  echo "# the one-pass backward (butd_attention_bwd_long_keys, the library's plan) next to the two-kernel walk, dropout 0.0 / 0.1; bf16: the bf16 entry points, two kernels | one pass:"
  BF16=1 python scratch/attn_longk_bench.py 2>/dev/null | grep " x "
  SHAPES=short python scratch/attn_longk_bench.py 2>/dev/null | grep " x "
  echo "# the bf16 entry points (bf16 LDS images, v_mfma_f32_16x16x32_bf16; round 4: fwd 96 / bwd 288 us at 1024 x 1024):"
  BF16=1 python scratch/attn_bench.py 2>/dev/null | grep "Lq="; } > $O/r06_attention_core.txt
python scratch/step_marks.py 20 > $O/r06_step_marks.txt 2>/dev/null
FINE=1 python scratch/step_marks.py 20 > $O/r06_step_marks_fine.txt 2>/dev/null
timeout 900 python bench.py > $O/r06_bench_default.json 2> $O/bench_default.err
tail -1 $O/r06_bench_default.json | cut -c1-300
