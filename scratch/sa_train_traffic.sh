#!/bin/bash
# HBM bytes and time of SA1 / SA2 forward + backward in training mode -> gpurun_out/<round>_sa_train_traffic.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/${ROUND:-r06}_sa_train_traffic.txt
echo "# set-abstraction levels in TRAINING mode at the bench size (8 x 50 000 points), forward + backward of fused_sa.sa_mlp_pool" > $OUT
echo "# time: graph replay; bytes: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, (2*FETCH_SIZE + WRITE_SIZE)*1024 (MI355X_MICROARCH.md), last of two steps" >> $OUT
for L in 1 2; do
  LEVEL=$L python scratch/sa_train_traffic.py 2>/dev/null | grep "training" >> $OUT
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/sa_$C
    LEVEL=$L timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/sa_$C -o c -- python scratch/sa_train_traffic.py pmc > /tmp/sa_$C.log 2>&1
  done
  LEVEL=$L python - >> $OUT <<'PY'
import csv, collections, os
def load(C):
    rows = [r for r in csv.DictReader(open(f"/tmp/sa_{C}/c_counter_collection.csv")) if r["Counter_Name"] == C]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows
f, w = load("FETCH_SIZE"), load("WRITE_SIZE")
names = [r["Kernel_Name"] for r in f]
# the last step = the second half of the dispatches after the setup kernels: take the last occurrence block
# one step = the distance between the two sa_pool_bwd_stats launches (one per step); the last step = the last `per` dispatches
marks = [i for i, n in enumerate(names) if "sa_pool_bwd_stats" in n]
per = marks[-1] - marks[-2]
last = len(names) - per
short = lambda n: (n.split("::")[-1] if "anonymous" in n else n).split("(")[0][:44]
tot = 0.0
print(f"SA{os.environ['LEVEL']}: per kernel of one forward + backward")
for rf, rw in zip(f[last:], w[last:]):
    mb = (2 * float(rf["Counter_Value"]) + float(rw["Counter_Value"])) * 1024 / 1e6
    tot += mb
    if mb > 1.0:
        print("  %-46s %9.1f MB" % (short(rf["Kernel_Name"]), mb))
print("  total %.2f GB" % (tot / 1e3))
PY
done
cat $OUT
