import csv, sys, collections
path, marker, pat = sys.argv[1], sys.argv[2], sys.argv[3]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]][:-7]
win = rows[marks[-2]:marks[-1]]
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
    if pat in r["Kernel_Name"]:
        key = (r["Kernel_Name"][:60], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg[key][0] += 1; agg[key][1] += d
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{d/1e3:9.1f} us total  x{c:<3d} avg {d/c/1e3:8.1f} us  grid {k[1]},{k[2]},{k[3]}  {k[0]}")
