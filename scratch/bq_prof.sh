cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
rm -rf /tmp/prof
BQ_ONLY_L1=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bq -- python scratch/bq_bench.py > /tmp/bq.log 2>&1
grep "n=\|b=" /tmp/bq.log
python - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/prof/bq_kernel_stats.csv')):
    n = r['Name'].replace('(anonymous namespace)::','')[:40]
    print(f"{n:42s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
