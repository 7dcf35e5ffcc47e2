import csv, sys
path, marker = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]][:-7]
win = rows[marks[-2]:marks[-1]]
t0 = int(win[0]["Start_Timestamp"])
lo, hi = float(sys.argv[3]), float(sys.argv[4])
for r in win:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
    if lo <= s / 1e3 <= hi:
        print(f"{s:9.1f} {e - s:8.1f} us q{r['Queue_Id']} {r['Kernel_Name'][:90]}")
