#!/bin/bash
# per-kernel durations of the attention core alone (scratch/attn_bench.py under rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python scratch/attn_bench.py 2>/dev/null
rm -rf /tmp/asplit
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/asplit -o a -- python scratch/attn_bench.py > /dev/null 2>&1
python - <<'PY'
import csv
for r in csv.DictReader(open("/tmp/asplit/a_kernel_stats.csv")):
    if "attn" in r["Name"]:
        print(f'{float(r["AverageNs"])/1e3:9.1f} us avg x{r["Calls"]:>5}  {r["Name"][:70]}')
PY
