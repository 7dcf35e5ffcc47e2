import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from butd_detr_amd import attention_blocks
from butd_detr_amd.bdetr import BeaUTyDETR
from tests.golden import text_stub, weights
from tests.golden.cases import PREFIXES, TRAIN_GRAD_KEYS, bdetr_inputs, train_loss, zero_dropout
g = np.load("tests/golden/bdetr_4096_train6.npz")
def rel(t, r):
    a = t.detach().float().cpu().numpy(); s = max(np.abs(r).max(), 1e-6)
    d = np.abs(a - r) / s
    return d.max(), (d > 2e-3).mean(), d.mean()
# HIP_ONLY=n: the fused path alone, n times (run-to-run spread of the atomics' summation order), worst max / mean over keys
CASES = ([("cuda", "hip")] * int(os.environ["HIP_ONLY"]) if os.environ.get("HIP_ONLY") else
         [(d, b) for d in (("cpu", "cuda") if torch.cuda.is_available() else ("cpu",)) for b in (("torch",) if d == "cpu" else ("torch", "hip"))])
for dev, backend in CASES:
  if True:
    attention_blocks.set_backend(backend)
    if dev == "cpu":
        from butd_detr_amd import pointnet2_utils
        from tests import oracle_ext
        pointnet2_utils._ext = oracle_ext
    else:
        from butd_detr_amd import pointnet2_utils, pointnet2_ext
        pointnet2_utils._ext = pointnet2_ext
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = BeaUTyDETR(num_class=256, num_obj_class=485, input_feature_dim=3, num_queries=82, num_decoder_layers=6,
                           self_position_embedding="loc_learned", contrastive_align_loss=True, butd=True, pointnet_ckpt=None,
                           self_attend=True, text_encoder_factory=text_stub.factory, class_embeddings_path="/nonexistent")
    weights.fill_(model, seed=15, skip_prefixes=("text_encoder.",))
    zero_dropout(model.to(dev).train())
    inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in bdetr_inputs().items()}
    ep = model(inp)
    train_loss(ep).backward()
    print("==", dev, backend)
    for k in ("seeds_obj_cls_logits", "proj_tokens", "last_proj_queries", "last_center", "0head_center", "proposal_center"):
        print("  out %-40s max %.2e" % (k, rel(ep[k], g[k])[0]))
    p = dict(model.named_parameters())
    for k in TRAIN_GRAD_KEYS:
        m, f, mean = rel(p[k].grad, g["g_" + k])
        print("  grad %-75s max %.2e  frac>2e-3 %.4f  mean %.2e" % (k, m, f, mean))
