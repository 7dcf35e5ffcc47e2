"""round 4: the flip list of tests/test_gpu_gradient_truth.py under different set-abstraction paths, one box."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import warnings; warnings.simplefilter("ignore")
    import torch
    from butd_detr_amd import attention_blocks
    from tests import grad_truth
    grad_truth.FIXED.clear()
    truth, _ = grad_truth.run("cpu", torch.float64, "torch")
    torch32, _ = grad_truth.run("cuda", torch.float32, "torch")
    hip32, _ = grad_truth.run("cuda", torch.float32, "hip")
    attention_blocks.set_backend("torch")
    top = max(float(t.abs().max()) for t in truth.values())
    flips = []
    for n, t in truth.items():
        sc = float(t.abs().max())
        if sc < 1e-6 * top:
            continue
        eh, et = (hip32[n] - t).abs() / sc, (torch32[n] - t).abs() / sc
        if float(eh.mean()) > 3 * float(et.mean()) + 5e-4 or float(eh.max()) > max(3 * float(et.max()), 6e-3):
            flips.append((n, round(float(eh.max()), 5), round(float(et.max()), 5), round(float(eh.mean()), 6), round(float(et.mean()), 6)))
    print("FLIPS", json.dumps(flips))
else:
    for tag, env in (("new", {}), ("old", {"BUTD_AB": "sa_last_bwd=0,sa_first_bwd=0"}),
                     ("new bwd, old fwd", {"BUTD_AB": "sa_last_fwd=0"})):
        out = subprocess.run([sys.executable, __file__, "child"], env={**os.environ, **env}, capture_output=True, text=True).stdout
        line = [l for l in out.splitlines() if l.startswith("FLIPS")]
        fl = json.loads(line[0][6:]) if line else None
        print(f"== {tag}: {None if fl is None else len(fl)} tensors outside the 3x band")
        for f in fl or []:
            print("    %-62s max %.5f (torch %.5f)  mean %.6f (torch %.6f)" % tuple(f))
