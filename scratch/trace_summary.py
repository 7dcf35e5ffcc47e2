"""Summarise ONE steady-state step from a rocprofv3 kernel trace: window = [start of the last-but-one
marker kernel, start of the last marker kernel)."""
import csv, sys, collections
path, marker = sys.argv[1], sys.argv[2]
skip_last = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if skip_last:
    marks = marks[:-skip_last]
a, b = marks[-2], marks[-1]
win = rows[a:b]
t0, t1 = int(win[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0])
busy = 0
for r in win:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[r["Kernel_Name"]][0] += 1
    agg[r["Kernel_Name"]][1] += d
    busy += d
print(f"step wall {(t1 - t0) / 1e6:.3f} ms, kernel busy {busy / 1e6:.3f} ms, {len(win)} launches")
for name, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{d / 1e6:9.3f} ms {100 * d / busy:5.1f}% x{c:<5d} {name[:150]}")

# category totals
cats = collections.defaultdict(lambda: [0, 0])
for name, (c, d) in agg.items():
    k = ("hand-written HIP" if "anonymous namespace)::" in name and "at::native" not in name else
         "hipBLASLt/rocBLAS" if name.startswith("Cijk") else "runtime copy/fill" if "rocclr" in name else
         "torch flash attention" if "attn_fwd" == name or "attn_bwd" in name and "anonymous" not in name else "torch elementwise/other")
    cats[k][0] += c; cats[k][1] += d
print("category totals:")
for k, (c, d) in sorted(cats.items(), key=lambda kv: -kv[1][1]):
    print(f"{d / 1e6:9.3f} ms {100 * d / busy:5.1f}% x{c:<5d} {k}")
