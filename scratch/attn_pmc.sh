cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/apmc
timeout 300 rocprofv3 --kernel-trace --pmc ${PMC:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU} --output-format csv -d /tmp/apmc -o a -- python scratch/attn_pmc.py > /tmp/apmc.log 2>&1
tail -2 /tmp/apmc.log
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('/tmp/apmc/a_counter_collection.csv')))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r['Kernel_Name']
    if 'attn_' not in n: continue
    key = 'attn_bwd_dq' if 'attn_bwd_dq' in n else 'attn_bwd_dkv' if 'attn_bwd_dkv' in n else 'attn_fwd'
    agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k)
    for c, v in d.items():
        print(f"   {c:28s} {sum(v)/len(v):14.0f}")
PY
