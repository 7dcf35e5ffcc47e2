#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export BUTD_AB=side_graphs=0
for m in none60 dummy1; do
rm -rf /tmp/pg_$m
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pg_$m -o t -- python scratch/side_cost.py $m 2>&1 | grep 'ms / step'
python scratch/r6_around.py /tmp/pg_$m/t_kernel_trace.csv 12 30
done
