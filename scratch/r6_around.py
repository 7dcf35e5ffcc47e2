"""round 6: the launches of ALL queues around a step boundary (the optimizer's kernel) of a kernel trace"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows)
idx = [i for i, k in enumerate(ks) if "adamw_flat_kernel" in k[3]]
i = idx[-6]
t0 = ks[i][0]
last_end = 0
for k in ks[i - int(sys.argv[2]):i + int(sys.argv[3])]:
    idle = (k[0] - last_end) / 1e3 if last_end else 0.0      # time since ANY queue last finished a kernel
    last_end = max(last_end, k[1])
    print(f"{(k[0]-t0)/1e3:9.1f} us  dur {(k[1]-k[0])/1e3:7.1f}  q{k[2]}  chip idle before {idle:7.1f}  {k[3][:80]}")
