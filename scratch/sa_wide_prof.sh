cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/swp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/swp -o a -- python scratch/sa_wide_bench.py > /tmp/swp.log 2>&1
grep wide= /tmp/swp.log
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('/tmp/swp/a_kernel_trace.csv')))
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if not any(k in n for k in ('gemm_kernel', 'sa_mid_wide', 'sa_dz_mid', 'sa_last_reduce', 'sa_mid_wide_finish', 'sa_last_sparse', 'sa_last_mfma', 'sa_last_fwd', 'sa_gather_rows')):
        continue
    import re
    short = re.search(r'::(\w+)', n).group(1) + (n[n.find('<'):n.find('>') + 1] if 'gemm_kernel' in n else '')
    agg[(short, r['Grid_Size_X'], r['Workgroup_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(agg.items()):
    print(f"{k[0][:48]:48s} grid {k[1]:>9s} wg {k[2]:>4s}  calls {len(v):5d}  avg {sum(v)/len(v):8.1f} us")
PY
