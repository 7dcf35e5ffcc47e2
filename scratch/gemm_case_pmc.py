"""Launches every case of scratch/gemm_cases.py three times (built-in tile choice), in order, so that a
rocprofv3 --pmc run can be sliced per case: rows of gemm_kernel dispatches come in groups of three."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TILES"] = "0x0"
import importlib.util
spec = importlib.util.spec_from_file_location("gc", os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_cases.py"))
src = open(spec.origin).read().split("only = os.environ.get")[0]      # definitions only, no timing loop
ns = {"__file__": spec.origin}
exec(compile(src, spec.origin, "exec"), ns)
torch, fa, keep = ns["torch"], ns["fa"], ns["keep"]
meta = []
for name, mk in ns["CASES"]:
    keep.clear()
    probs = mk()
    flops = sum(2.0 * p.M * p.N * p.K for p in probs)
    # algorithmic bytes: every operand and result once
    byts = sum(4.0 * (p.M * p.K + p.N * p.K + p.M * p.N) for p in probs)
    for _ in range(3):
        fa._gemm(probs, keep[0])
    torch.cuda.synchronize()
    meta.append({"case": name, "flops": flops, "algorithmic_bytes": byts})
    torch.cuda.empty_cache()
json.dump(meta, open(os.environ.get("CASE_META", "/tmp/case_meta.json"), "w"))
