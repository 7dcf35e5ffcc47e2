"""round 4: error of selected gradients against the float64 truth (tests/grad_truth.py), fused path vs stock torch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings; warnings.simplefilter("ignore")
import torch
from butd_detr_amd import attention_blocks
from tests import grad_truth
grad_truth.FIXED.clear()
truth, _ = grad_truth.run("cpu", torch.float64, "torch")
torch32, _ = grad_truth.run("cuda", torch.float32, "torch")
hip32, _ = grad_truth.run("cuda", torch.float32, "hip")
attention_blocks.set_backend("torch")
for n in truth:
    if n.startswith("backbone_net.sa1"):
        sc = float(truth[n].abs().max())
        if sc == 0: continue
        eh, et = (hip32[n] - truth[n]).abs() / sc, (torch32[n] - truth[n]).abs() / sc
        print("%-50s fused max %.2e mean %.2e   torch max %.2e mean %.2e" % (n, float(eh.max()), float(eh.mean()), float(et.max()), float(et.mean())))
