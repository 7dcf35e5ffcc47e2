import csv, sys, collections
path, marker = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(list(rows[0].keys()))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]][:-7]
win = rows[marks[-2]:marks[-1]]
t0, t1 = int(win[0]["Start_Timestamp"]), int(rows[marks[-1]]["Start_Timestamp"])
key = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[r.get(key, "?")][0] += 1; agg[r.get(key, "?")][1] += d
print("wall %.3f ms" % ((t1 - t0) / 1e6))
for k, (c, d) in agg.items():
    print(key, k, "launches", c, "busy %.3f ms" % (d / 1e6))
# union of busy intervals (any queue)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in win)
cur_s, cur_e, tot = iv[0][0], iv[0][1], 0
for s, e in iv[1:]:
    if s > cur_e:
        tot += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
tot += cur_e - cur_s
print("union busy %.3f ms -> idle %.3f ms" % (tot / 1e6, (t1 - t0 - tot) / 1e6))
