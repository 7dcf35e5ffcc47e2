"""round 6: main-queue launch gaps of a kernel trace: python r6_gap_hist.py trace.csv -> per queue: launches, busy, gaps (histogram)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
main = max(byq, key=lambda q: len(byq[q]))
ks = sorted(byq[main])
ks = ks[len(ks) // 2:]                      # the second half: steady replays
span = ks[-1][1] - ks[0][0]
busy = sum(e - s for s, e, _ in ks)
gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
print(f"main queue {main}: {len(ks)} launches, span {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms, gaps {sum(gaps)/1e6:.2f} ms")
edges = [0, 500, 1000, 2000, 3000, 4000, 6000, 10000, 50000, 10**9]
h = collections.Counter()
tot = collections.Counter()
for g in gaps:
    for lo, hi in zip(edges, edges[1:]):
        if lo <= g < hi or (g < 0 and lo == 0):
            h[(lo, hi)] += 1; tot[(lo, hi)] += max(g, 0); break
for lo, hi in zip(edges, edges[1:]):
    print(f"  gap {lo/1e3:5.1f} .. {hi/1e3:8.1f} us: {h[(lo,hi)]:6d} launches  {tot[(lo,hi)]/1e6:7.3f} ms   per launch {tot[(lo,hi)]/max(h[(lo,hi)],1)/1e3:6.2f} us")
print("  mean kernel duration %.2f us" % (busy / len(ks) / 1e3))
for q, v in byq.items():
    if q != main: print(f"  queue {q}: {len(v)} launches")
