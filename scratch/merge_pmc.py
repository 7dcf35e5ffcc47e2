"""gpurun_out/pmc/{FETCH_SIZE,WRITE_SIZE}.json -> profiles/r0N_pmc.json (HBM bytes per launch, corrected as
MI355X_MICROARCH.md's HBM section prescribes for gfx950)."""
import json, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
f = json.load(open(f"{src}/FETCH_SIZE.json"))
w = json.load(open(f"{src}/WRITE_SIZE.json"))
out = {"_provenance": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python bench.py "
                      "--steps 2 --warmup 1 --no-cpu-baseline on MI355X (scratch/pmc.sh, merged by scratch/merge_pmc.py); "
                      "averages over every launch of the kernel in the run",
       "_correction": "hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts 128-B requests at 64 B "
                      "for wide coalesced reads (MI355X_MICROARCH.md section HBM); WRITE_SIZE is uncalibrated"}
for k in f:
    if k not in w:
        continue
    fk, wk = f[k]["avg_FETCH_SIZE"], w[k]["avg_WRITE_SIZE"]
    out[k] = {"launches": f[k]["launches"], "avg_FETCH_SIZE_kb": round(fk, 1), "avg_WRITE_SIZE_kb": round(wk, 1),
              "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
json.dump(out, open("profiles/r02_pmc.json" if len(sys.argv) < 3 else sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
