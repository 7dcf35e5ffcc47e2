"""main-queue kernels of one steady-state step shorter than LIMIT us (graph-node floor ~4.6 us), by name: count, total,
and the 1-ms windows they fall into -- the launches that are pure overhead.  small_kernels.py trace.csv marker skip [limit]"""
import csv, sys, collections, re
path, marker, skip_last = sys.argv[1], sys.argv[2], int(sys.argv[3])
LIMIT = float(sys.argv[4]) if len(sys.argv) > 4 else 7.0
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
def short(n):
    n = n.replace("void ", "").replace("at::native::", "").replace("(anonymous namespace)::", "")
    f = re.search(r"(\w+Functor\w*|direct_copy\w*|sum_functor|\w+Op)\b", n)
    head = re.match(r"[\w:]+", n).group(0)
    return (head + (":" + f.group(1) if f and f.group(1) not in head else ""))[:64]
agg = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
tot = [0, 0.0]
for r in m:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d < LIMIT:
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1; a[1] += d; a[2][(int(r["Start_Timestamp"]) - t0) // 1000000] += 1
        tot[0] += 1; tot[1] += d
print(f"main queue: {len(m)} launches, {tot[0]} shorter than {LIMIT} us = {tot[1] / 1e3:.3f} ms")
for k, (n, d, w) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:4d} x {d / n:4.1f} us = {d:6.0f} us  {k:64s} windows(ms): " + " ".join(f"{a}:{c}" for a, c in sorted(w.items())))
