"""Round-4 diagnosis of tests/test_gpu_gradient_truth.py failing on SOME boxes with
('prediction_heads.3.size_pred_head.net.1.weight', 3.1e-3, 5.6e-6): is it a discrete ReLU decision of one channel
(the test's "flip"), and does the encoder position hand-off (BUTD_ENC_POS_HANDOFF) change any forward value?
Prints the box, a checksum of the fused forward, the error of that tensor, and -- replaying the size head of decoder
layer 3 in float64 on the features each run handed it -- the gates that differ from the float64 truth."""
import os, sys, torch, warnings, hashlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
warnings.simplefilter("ignore")
from butd_detr_amd import attention_blocks
from tests import grad_truth

p = torch.cuda.get_device_properties(0)
print("box:", p.name, p.multi_processor_count, "CUs", getattr(p, "clock_rate", "?"), "kHz", torch.version.hip)
feats = {}
orig_build = grad_truth.build
def build():
    m = orig_build()
    m.prediction_heads[3].register_forward_pre_hook(
        lambda mod, args, kwargs: feats.__setitem__("cur", (kwargs.get("features_pm") if kwargs.get("features_pm") is not None
                                                            else args[0].transpose(1, 2)).detach().double().cpu()),
        with_kwargs=True)
    feats["model"] = m
    return m
grad_truth.build = build
grad_truth.FIXED.clear()
truth, ept = grad_truth.run("cpu", torch.float64, "torch")
f_truth, head = feats["cur"], feats["model"].prediction_heads[3].size_pred_head.double().cpu()
runs = {}
for h in ("1", "0"):
    os.environ["BUTD_ENC_POS_HANDOFF"] = h
    g, ep = grad_truth.run("cuda", torch.float32, "hip")
    runs[h] = (g, ep, feats["cur"])
attention_blocks.set_backend("torch")
def checksum(ep):
    hsh = hashlib.sha1()
    for k in sorted(ep):
        hsh.update(ep[k].numpy().tobytes())
    return hsh.hexdigest()[:12]
n = "prediction_heads.3.size_pred_head.net.1.weight"
for h in ("1", "0"):
    g, ep, f = runs[h]
    e = (g[n] - truth[n]).abs() / float(truth[n].abs().max())
    print("handoff", h, "forward checksum", checksum(ep), " %s: mean err %.3e max %.3e" % (n, float(e.mean()), float(e.max())))
print("forward tensors equal handoff on/off:", all(torch.equal(runs["1"][1][k], runs["0"][1][k]) for k in runs["1"][1]))

def gates(f):       # (B, Q, C) -> pre-activations of the two ReLUs (train-mode BatchNorm over B*Q rows), float64
    x = f.reshape(-1, f.shape[-1])
    out = []
    net = head.net
    for conv, bn in ((net[0], net[1]), (net[4], net[5])):
        z = x @ conv.weight[:, :, 0].T
        z = (z - z.mean(0)) / torch.sqrt(z.var(0, unbiased=False) + bn.eps) * bn.weight + bn.bias
        out.append(z)
        x = torch.relu(z)
    return out
zt = gates(f_truth)
zh = gates(runs["1"][2])
print("features handed to head 3: max |fused - truth| = %.3e of %.3e" % (float((runs["1"][2] - f_truth).abs().max()), float(f_truth.abs().max())))
for l, (a, b) in enumerate(zip(zt, zh)):
    differ = ((a > 0) != (b > 0)).nonzero()
    print("layer %d: %d of %d gates decide differently" % (l + 1, len(differ), a.numel()))
    for r, c in differ[:8].tolist():
        print("    row %d (scene %d, query %d) channel %d: truth pre-activation %+.3e, fused features give %+.3e (column scale %.2f)"
              % (r, r // f_truth.shape[1], r % f_truth.shape[1], c, float(a[r, c]), float(b[r, c]), float(a[:, c].abs().mean())))
