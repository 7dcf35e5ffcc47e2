"""GPU-idle intervals (no queue busy) of one steady-state step of a kernel trace"""
import csv, sys
path, marker = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]][:-7]
win = rows[marks[-3]:marks[-2]]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in win)
t0 = ev[0][0]; ce = ev[0][1]; last = ev[0][2]; gaps = []
for s, e, n in ev[1:]:
    if s > ce: gaps.append((s - ce, (ce - t0) / 1e6, last[:50], n[:50]))
    if e > ce: ce = e; last = n
print("step wall %.3f ms, launches %d, idle %.3f ms in %d gaps; gaps > 10 us:" % ((ev[-1][1] - t0) / 1e6, len(ev), sum(g[0] for g in gaps) / 1e6, len(gaps)))
for g in sorted(gaps, reverse=True)[:10]:
    if g[0] > 10000: print("%8.1f us at %7.3f ms  after [%s]  before [%s]" % (g[0] / 1e3, g[1], g[2], g[3]))
