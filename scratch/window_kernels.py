"""kernels of the main queue between two offsets (ms) of one steady-state step of a rocprofv3 kernel trace, in order:
window_kernels.py trace.csv marker skip_last from_ms to_ms"""
import csv, sys, collections, re
path, marker, skip_last, lo, hi = sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5])
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if skip_last:
    marks = marks[:-skip_last]
win = rows[marks[-2]:marks[-1]]
byq = collections.defaultdict(list)
for r in win:
    byq[r["Queue_Id"]].append(r)
mainq = max(byq, key=lambda q: len(byq[q]))
m = byq[mainq]
t0 = int(m[0]["Start_Timestamp"])
prev_end = None
import os
if os.environ.get("ALLQ") == "1":          # every queue, tagged (the main queue = *)
    m = sorted(win, key=lambda r: int(r["Start_Timestamp"]))
for r in m:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    if lo <= s <= hi:
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")
        gap = (s - prev_end) * 1e3 if prev_end is not None else 0.0
        tag = ("*" if r["Queue_Id"] == mainq else " ") + r["Queue_Id"]
        print(f"{s:8.3f} ms  {1e3 * (e - s):7.1f} us  gap {gap:5.1f}  q{tag}  grid {r['Grid_Size_X']:>8s}  {n[:150]}")
    prev_end = e
