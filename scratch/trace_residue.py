"""one steady-state step of a kernel trace: the main queue's busy time and gaps, and what the stock-torch launches
(at::native / rocclr copies) cost once the gap that follows each launch is charged to it."""
import csv, sys, collections
path, marker = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]][:-7]
win = rows[marks[-2]:marks[-1]]
byq = collections.defaultdict(list)
for r in win:
    byq[r["Queue_Id"]].append(r)
mainq = max(byq, key=lambda q: len(byq[q]))
m = byq[mainq]
wall = int(m[-1]["End_Timestamp"]) - int(m[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in m)
print(f"main queue: {len(m)} launches, wall {wall/1e6:.3f} ms, busy {busy/1e6:.3f} ms, idle {(wall-busy)/1e6:.3f} ms "
      f"({(wall-busy)/len(m)/1e3:.2f} us per launch); other queues: " + ", ".join(f"{len(v)}" for q, v in byq.items() if q != mainq))
fam = collections.defaultdict(lambda: [0, 0, 0])
for a, b in zip(m[:-1], m[1:]):
    n = a["Kernel_Name"]
    if "at::native" in n or "rocclr" in n or "at::cuda" in n:
        key = "torch: " + n.split("<")[0].replace("void at::native::", "")[:40] + ("/" + n.split("at::native::")[2].split("<")[0][:30] if n.count("at::native::") > 1 else "")
    elif "Cijk" in n:
        key = "hipBLASLt"
    else:
        key = "own: " + n.split("::")[-1].split("(")[0].split("<")[0][:40]
    d = int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
    g = max(0, int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
    f = fam[key]; f[0] += 1; f[1] += d; f[2] += g
print("%-75s %5s %9s %9s" % ("family", "n", "busy ms", "gap ms"))
for k, (n, d, g) in sorted(fam.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print("%-75s %5d %9.3f %9.3f" % (k, n, d / 1e6, g / 1e6))
tn = sum(v[0] for k, v in fam.items() if k.startswith("torch")); td = sum(v[1] for k, v in fam.items() if k.startswith("torch")); tg = sum(v[2] for k, v in fam.items() if k.startswith("torch"))
print(f"stock torch on the main queue: {tn} launches, {td/1e6:.3f} ms busy + {tg/1e6:.3f} ms of gaps after them")
