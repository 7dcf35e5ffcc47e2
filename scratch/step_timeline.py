"""one steady-state step of a rocprofv3 kernel trace as a timeline of the main queue: per 0.5 ms bin the busy time,
the number of launches and the dominant kernels; plus every gap > GAP us with the kernels around it."""
import csv, sys, collections
path, marker = sys.argv[1], sys.argv[2]
skip_last = int(sys.argv[3]) if len(sys.argv) > 3 else 0
GAP = float(sys.argv[4]) if len(sys.argv) > 4 else 10.0
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
short = lambda n: (n.split("::")[-1] if "anonymous namespace)::" in n and "at::native" not in n else n.replace("void at::native::", "")).split("(")[0][:60]
t0 = int(m[0]["Start_Timestamp"])
print(f"main queue {mainq}: {len(m)} launches; other queues: " + ", ".join(f"{q}:{len(v)}" for q, v in byq.items() if q != mainq))
for q, v in byq.items():
    if q != mainq:
        print(f"  queue {q}: {(int(v[0]['Start_Timestamp']) - t0) / 1e6:.2f} .. {(int(v[-1]['End_Timestamp']) - t0) / 1e6:.2f} ms, busy "
              f"{sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in v) / 1e6:.2f} ms; first {short(v[0]['Kernel_Name'])}, last {short(v[-1]['Kernel_Name'])}")
bins = collections.defaultdict(lambda: [0, 0, collections.Counter()])
for r in m:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    b = bins[s // 500000]
    b[0] += e - s; b[1] += 1; b[2][short(r["Kernel_Name"])] += e - s
print("bin(ms)  busy%  launches  top kernels")
for k in sorted(bins):
    busy, n, c = bins[k]
    print(f"{k * 0.5:6.1f}  {busy / 5000:5.0f}  {n:5d}   " + ", ".join(f"{a}:{d / 1e3:.0f}us" for a, d in c.most_common(3)))
print(f"gaps > {GAP} us on the main queue:")
for a, b in zip(m[:-1], m[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if g > GAP:
        print(f"  at {(int(a['End_Timestamp']) - t0) / 1e6:7.3f} ms  gap {g:7.1f} us  after {short(a['Kernel_Name'])}  before {short(b['Kernel_Name'])}")
