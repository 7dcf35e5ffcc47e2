import csv, sys, collections
path, marker = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]][:-7]
win = rows[marks[-2]:marks[-1]]
t0 = int(win[0]["Start_Timestamp"])
byq = collections.defaultdict(list)
for r in win:
    byq[r["Queue_Id"]].append(r)
mainq = max(byq, key=lambda q: len(byq[q]))
print("main queue", mainq, {q: len(v) for q, v in byq.items()})
for q, v in byq.items():
    s = int(v[0]["Start_Timestamp"]) - t0; e = max(int(r["End_Timestamp"]) for r in v) - t0
    print(f"queue {q}: first start {s/1e6:.3f} ms, last end {e/1e6:.3f} ms, first kernel {v[0]['Kernel_Name'][:60]}, last {v[-1]['Kernel_Name'][:60]}")
m = byq[mainq]
prev_end = int(m[0]["End_Timestamp"])
gaps = []
for a, b in zip(m[:-1], m[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g > 20000:
        gaps.append((g, int(a["End_Timestamp"]) - t0, a["Kernel_Name"][:50], b["Kernel_Name"][:50]))
tot = sum(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(m[:-1], m[1:]) if int(b["Start_Timestamp"]) > int(a["End_Timestamp"]))
print("main queue total gap %.3f ms; gaps > 20us:" % (tot / 1e6))
for g, at, a, b in sorted(gaps, reverse=True)[:25]:
    print(f"  {g/1e3:8.1f} us at {at/1e6:7.3f} ms  after [{a}] before [{b}]")
