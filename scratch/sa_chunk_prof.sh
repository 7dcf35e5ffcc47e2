cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in 2048 4096; do
  rm -rf /tmp/prof$c
  BUTD_SA_CHUNK=$c timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof$c -o b -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --criterion surrogate > /tmp/b$c.log 2>&1
  echo "== chunk $c"
  python - <<PY
import csv
for r in csv.DictReader(open('/tmp/prof$c/b_kernel_stats.csv')):
    n = r['Name'].replace('(anonymous namespace)::','')
    if n.startswith('sa_'):
        print(f"{n[:28]:30s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e6:7.3f} ms")
PY
done
