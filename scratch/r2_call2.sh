mkdir -p gpurun_out/r2c2
export CASES="fwd 3x(8192"
export TILES="64x64,128x64,64x96,32x32"
for m in 0 1 2 4 8 16 14 30 15; do
  if [ $m = 0 ]; then unset BUTD_HIP_LIB; else export BUTD_HIP_LIB=$PWD/scratch/exp/libabl_$m.so; fi
  echo "== ablation mask $m"
  timeout 300 python scratch/gemm_cases.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r2c2/abl.txt 2>&1
cat gpurun_out/r2c2/abl.txt
unset BUTD_HIP_LIB
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for t in 64x64 128x64; do
rm -rf /tmp/gpmc
TILE=$t timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/gpmc -o a -- python scratch/gemm_pmc.py > /tmp/gpmc.log 2>&1
tail -2 /tmp/gpmc.log
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('/tmp/gpmc/a_counter_collection.csv')))
agg = collections.defaultdict(list)
for r in rows:
    if 'gemm_kernel' not in r['Kernel_Name']: continue
    agg[r['Counter_Name']].append(float(r['Counter_Value']))
for c, v in agg.items():
    print(f"   {c:28s} {sum(v)/len(v):14.0f}  (n={len(v)})")
PY
rm -rf /tmp/gpmc
TILE=$t timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d /tmp/gpmc -o a -- python scratch/gemm_pmc.py > /tmp/gpmc.log 2>&1
tail -2 /tmp/gpmc.log
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('/tmp/gpmc/a_counter_collection.csv')))
agg = collections.defaultdict(list)
for r in rows:
    if 'gemm_kernel' not in r['Kernel_Name']: continue
    agg[r['Counter_Name']].append(float(r['Counter_Value']))
for c, v in agg.items():
    print(f"   {c:28s} {sum(v)/len(v):14.0f}  (n={len(v)})")
PY
done > gpurun_out/r2c2/pmc.txt 2>&1
cat gpurun_out/r2c2/pmc.txt
