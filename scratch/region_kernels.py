"""Kernels between two timeline marks: rocprofv3 --kernel-trace of scratch/step_marks.py; the mark kernels
(timeline_mark_kernel) segment the main queue's trace of the LAST replay.  usage: region_kernels.py trace.csv names.txt"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [l.rstrip("\n") for l in open(sys.argv[2])]
marks = [i for i, r in enumerate(rows) if "timeline_mark_kernel" in r["Kernel_Name"]]
n = len(names)
last = marks[-n:]                       # the last replay's marks
qcount = collections.Counter(r["Queue_Id"] for r in rows[last[0]:last[-1]])
mainq = qcount.most_common(1)[0][0]
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 200
for j in range(n - 1):
    a, b = last[j], last[j + 1]
    t0, t1 = int(rows[a]["End_Timestamp"]), int(rows[b]["Start_Timestamp"])
    dur = (t1 - t0) / 1e3
    if dur < thr:
        continue
    seg = [r for r in rows[a + 1:b] if r["Queue_Id"] == mainq and "timeline_mark" not in r["Kernel_Name"]]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3
    print(f"== {names[j]} -> {names[j + 1]}: {dur:.0f} us, {len(seg)} launches on the main queue, busy {busy:.0f} us")
    agg = collections.OrderedDict()
    for r in seg:
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70] or r["Kernel_Name"][:70]
        a_ = agg.setdefault(k, [0, 0.0]); a_[0] += 1; a_[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f"     {d:8.1f} us x{c:<3d} {k}")
