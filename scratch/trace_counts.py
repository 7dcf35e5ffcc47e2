import csv, sys, collections
path, marker = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]][:-7]
win = rows[marks[-2]:marks[-1]]
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[r["Kernel_Name"]][0] += 1; agg[r["Kernel_Name"]][1] += d
print(len(win), "launches")
for name, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"x{c:<4d} {d/1e3:8.1f} us  {name[:130]}")
