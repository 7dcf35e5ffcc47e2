"""one steady-state step of a rocprofv3 kernel trace: the kernels on each queue that are NOT this library's (torch
element-wise / copies / fills / reductions, hipBLASLt), per queue and per 1 ms window of the main queue's clock --
what is still a stock launch, where in the step, and how long it runs."""
import csv, sys, collections, re
path, marker = sys.argv[1], sys.argv[2]
skip_last = int(sys.argv[3]) if len(sys.argv) > 3 else 0
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
t0 = int(byq[mainq][0]["Start_Timestamp"])
ours = lambda n: "(anonymous namespace)::" in n and "at::native" not in n
def short(n):
    n = n.replace("void at::native::", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([\w:]+)<?", n)
    head = m.group(1) if m else n[:40]
    f = re.search(r"(\w+Functor|\w+_kernel_cuda|direct_copy\w*|\w+Op)\b", n)
    return (head + (":" + f.group(1) if f and f.group(1) not in head else ""))[:70]
for q, v in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    stock = [r for r in v if not ours(r["Kernel_Name"])]
    dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print(f"queue {q}{' (main)' if q == mainq else ''}: {len(v)} launches, {len(stock)} stock ones = {sum(map(dur, stock)) / 1e6:.3f} ms")
    if q != mainq:
        c = collections.Counter()
        for r in stock:
            c[short(r["Kernel_Name"])] += 1
        print("    " + ", ".join(f"{k} x{n}" for k, n in c.most_common(8)))
        continue
    wins = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0]))
    for r in stock:
        w = (int(r["Start_Timestamp"]) - t0) // 1000000
        e = wins[w][short(r["Kernel_Name"])]
        e[0] += 1; e[1] += dur(r)
    for w in sorted(wins):
        tot = sum(e[1] for e in wins[w].values())
        print(f"  {w:3d}-{w + 1:<3d} ms: {sum(e[0] for e in wins[w].values()):3d} launches {tot / 1e3:6.0f} us   " +
              ", ".join(f"{k} x{e[0]} {e[1] / 1e3:.0f}us" for k, e in sorted(wins[w].items(), key=lambda kv: -kv[1][1])[:6]))
