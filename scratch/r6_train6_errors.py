"""round 6: per-tensor gradient errors of the train-mode 3 + 6-layer model against the reference's golden vectors
(tests/golden/bdetr_4096_train6.npz), fused backend, for the BUTD_AB setting of the environment."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.golden import text_stub, weights
from tests.golden.cases import bdetr_inputs, train_loss, zero_dropout, TRAIN_GRAD_KEYS
from butd_detr_amd import attention_blocks
from butd_detr_amd.bdetr import BeaUTyDETR
attention_blocks.set_backend(os.environ.get("BACKEND", "hip"))
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bdetr_4096_train6.npz"))
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    model = BeaUTyDETR(num_class=256, num_obj_class=485, input_feature_dim=3, num_queries=82, num_decoder_layers=6,
                       self_position_embedding="loc_learned", contrastive_align_loss=True, butd=True, pointnet_ckpt=None,
                       self_attend=True, text_encoder_factory=text_stub.factory,
                       class_embeddings_path="/nonexistent/class_embeddings3d.npy")
weights.fill_(model, seed=15, skip_prefixes=("text_encoder.",))
zero_dropout(model.cuda().train())
inp = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in bdetr_inputs().items()}
ep = model(inp)
train_loss(ep).backward()
p = dict(model.named_parameters())
print(f"BUTD_AB={os.environ.get('BUTD_AB', '')!r} backend={attention_blocks.get_backend()}")
for k in TRAIN_GRAD_KEYS:
    ref = g["g_" + k]
    a = p[k].grad.detach().float().cpu().numpy()
    e = np.abs(a - ref) / max(float(np.abs(ref).max()), 1e-6)
    print(f"  {e.max():.3e} max  {e.mean():.3e} mean   {k}")
