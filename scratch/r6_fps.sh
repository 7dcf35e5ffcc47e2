#!/bin/bash
# round 6: the FPS prefix check -- parity tests, then per-kernel durations of levels 2-4 (8 scenes) under rocprofv3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fps_prefix.py tests/test_gpu_pointnet2_parity.py -x -q -s -m gpu > $O/fps_tests.log 2>&1; grep -n 'FPS levels\|passed\|failed\|Error\|error' $O/fps_tests.log | head -20
rm -rf /tmp/prof_fps
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fps -o fps -- python -m pytest tests/test_gpu_fps_prefix.py -q -s -m gpu -k levels_2_to_4_time > /dev/null 2>&1
python - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/prof_fps/fps_kernel_stats.csv')):
    if 'fps' in r['Name']: print('%6d calls  avg %8.1f us  min %8.1f  max %8.1f  %s' % (int(r['Calls']), float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, r['Name'][:90]))
PY
