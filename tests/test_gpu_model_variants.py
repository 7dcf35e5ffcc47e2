"""-m gpu: the constructor variants of BeaUTyDETR (bdetr.py:46-52) run forward + backward on the fused gfx950
path and agree with the stock-torch maths of the same modules: no detected-box stream, no / xyz-only
learned query position embedding, no contrastive heads, no visual/language self-attention."""
import copy
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

VARIANTS = [
    dict(butd=False),
    dict(self_position_embedding="none"),
    dict(self_position_embedding="xyz_learned"),
    dict(contrastive_align_loss=False),
    dict(self_attend=False),
]


@pytest.mark.parametrize("kw", VARIANTS, ids=lambda k: ",".join(f"{a}={b}" for a, b in k.items()))
def test_variant_matches_torch_backend(kw):
    from butd_detr_amd import attention_blocks
    from butd_detr_amd.bdetr import BeaUTyDETR
    from butd_detr_amd.offline_text import offline_factory
    from butd_detr_amd.train_step import synthetic_batch
    cfg = dict(num_class=256, num_obj_class=485, input_feature_dim=3, num_queries=64, num_decoder_layers=2,
               num_encoder_layers=1, self_position_embedding="loc_learned", contrastive_align_loss=True,
               butd=True, self_attend=True, text_encoder_factory=offline_factory(0))
    cfg.update(kw)
    try:
        torch.manual_seed(1)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = BeaUTyDETR(**cfg).cuda().eval()
        fused = copy.deepcopy(ref)
        inputs, _ = synthetic_batch(2, torch.device("cuda", 0), seed=5, n_points=4096, tokens=24)
        res = {}
        for name, model, backend in (("torch", ref, "torch"), ("hip", fused, "hip")):
            attention_blocks.set_backend(backend)
            ep = model(inputs)
            loss = (ep["last_center"].square().sum() + ep["last_sem_cls_scores"].square().mean()
                    + ep["seeds_obj_cls_logits"].square().mean())
            if cfg["contrastive_align_loss"]:
                loss = loss + ep["last_proj_queries"].sum() + ep["proj_tokens"].square().sum()
            loss.backward()
            res[name] = (ep, float(loss.detach()))
        (ep_t, lt), (ep_h, lh) = res["torch"], res["hip"]
        assert set(ep_t.keys()) == set(ep_h.keys())
        assert abs(lh - lt) <= 5e-3 * max(abs(lt), 1.0), (lh, lt)
        for key in ("seed_features", "text_memory", "seeds_obj_cls_logits"):
            a, b = ep_h[key].detach().cpu().numpy(), ep_t[key].detach().cpu().numpy()
            np.testing.assert_allclose(a / max(np.abs(b).max(), 1e-6), b / max(np.abs(b).max(), 1e-6), rtol=0, atol=2e-3)
        # (a head whose output the loss does not read gets None from torch and exact zeros from the
        # grouped fused head: both mean "no gradient")
        none = lambda g: g is None or float(g.abs().max()) == 0.0
        missing = [n for (n, p), (_, q) in zip(fused.named_parameters(), ref.named_parameters())
                   if none(p.grad) != none(q.grad)]
        assert not missing, missing[:8]
    finally:
        attention_blocks.set_backend("torch")


def test_config0_one_cloud_one_query_eight_tokens():
    """BASELINE configs[0] as written: a single 4096-point random cloud, ONE query, an 8-token utterance -- the
    plumbing shape (degenerate tiles everywhere: Lq = 1 attention, a 1-row head, top-1 query selection).  Fused gfx950
    path vs the stock-torch maths of the same modules (eval: BatchNorm over one sample has no batch statistics), and
    a backward pass through both."""
    from butd_detr_amd import attention_blocks
    from butd_detr_amd.bdetr import BeaUTyDETR
    from butd_detr_amd.offline_text import offline_factory
    from butd_detr_amd.synthetic_scenes import detected_boxes, uniform_cloud
    try:
        torch.manual_seed(2)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = BeaUTyDETR(num_class=256, num_obj_class=485, input_feature_dim=3, num_queries=1,
                             num_decoder_layers=6, self_position_embedding="loc_learned", contrastive_align_loss=True,
                             butd=True, self_attend=True, text_encoder_factory=offline_factory(0)).cuda().eval()
        fused = copy.deepcopy(ref)
        dev = torch.device("cuda", 0)
        boxes, mask, cls = detected_boxes(1, seed=0, min_valid=20, max_valid=20)       # 132 slots, 20 valid
        inputs = {"point_clouds": torch.from_numpy(uniform_cloud(seed=0, n_points=4096, batch=1)).to(dev),
                  "text": ["the chair next to the brown table ."],
                  "det_boxes": torch.from_numpy(boxes).to(dev), "det_bbox_label_mask": torch.from_numpy(mask).to(dev),
                  "det_class_ids": torch.from_numpy(cls).to(dev)}
        res = {}
        for name, model, backend in (("torch", ref, "torch"), ("hip", fused, "hip")):
            attention_blocks.set_backend(backend)
            ep = model(inputs)
            (ep["last_center"].square().sum() + ep["last_sem_cls_scores"].square().mean()
             + ep["last_proj_queries"].sum() + ep["proj_tokens"].square().sum()).backward()
            res[name] = ep
        ep_t, ep_h = res["torch"], res["hip"]
        assert ep_h["text_feats"].shape[1] <= 12 and ep_h["last_center"].shape == (1, 1, 3)
        assert ep_h["query_points_sample_inds"].shape == (1, 1)
        assert torch.equal(ep_h["query_points_sample_inds"], ep_t["query_points_sample_inds"])
        assert torch.equal(ep_h["seed_inds"], ep_t["seed_inds"])
        for key in ("seed_features", "text_memory", "seeds_obj_cls_logits", "proposal_center", "last_center",
                    "last_pred_size", "last_sem_cls_scores", "last_proj_queries", "proj_tokens"):
            a, b = ep_h[key].detach().cpu().numpy(), ep_t[key].detach().cpu().numpy()
            np.testing.assert_allclose(a / max(np.abs(b).max(), 1e-6), b / max(np.abs(b).max(), 1e-6), rtol=0,
                                       atol=2e-3, err_msg=key)
        g_h = fused.decoder[5].cross_v.in_proj_weight.grad
        g_t = ref.decoder[5].cross_v.in_proj_weight.grad
        assert float((g_h - g_t).abs().max() / g_t.abs().max()) <= 5e-3
    finally:
        attention_blocks.set_backend("torch")
