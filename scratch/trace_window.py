"""last steady-state steps of a kernel trace -> compact csv (queue, start ns, end ns, name[:90]) for offline study"""
import csv, sys
path, marker, out = sys.argv[1], sys.argv[2], sys.argv[3]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
print("markers", len(marks), "rows", len(rows), "cols", list(rows[0].keys()))
marks = marks[:-7]
lo, hi = marks[-4], marks[-1]
t0 = int(rows[lo]["Start_Timestamp"])
with open(out, "w") as f:
    w = csv.writer(f)
    w.writerow(["queue", "stream", "start", "end", "name"])
    for r in rows[lo:hi]:
        w.writerow([r["Queue_Id"], r.get("Stream_Id", ""), int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Kernel_Name"][:90]])
