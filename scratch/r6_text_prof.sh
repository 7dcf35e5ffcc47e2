#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/pt; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o t -- python scratch/r6_text_tower.py > /dev/null 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/pt/t_kernel_stats.csv')))
for r in rows[:22]:
    print('%6d calls  avg %7.1f us  total %8.1f ms  %s'%(int(r['Calls']), float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Name'][:110]))
PY
