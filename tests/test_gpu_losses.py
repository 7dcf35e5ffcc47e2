"""-m gpu: the device-side Hungarian matching (butd_hungarian_match, include/butd_lsap.h) against scipy's
linear_sum_assignment -- the third-party routine HungarianMatcher.forward calls (models/losses.py:314-319) --
and the complete criterion on the GPU against vectors captured from the reference's models/losses.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.test_losses_cpu import GOLD, check_outputs, load_case  # noqa: E402


def scipy_match(cost_gq, valid):
    """Dense form of what the reference does with one scene: compact, solve (queries x targets), scatter."""
    from scipy.optimize import linear_sum_assignment
    match = -np.ones(valid.shape[0], dtype=np.int32)
    slots = np.nonzero(valid)[0]
    if slots.size:
        q, t = linear_sum_assignment(cost_gq[slots].T)
        match[slots[t]] = q
    return match


def run_match(cost, valid):
    from butd_detr_amd import losses as L
    match, status = L.hungarian_match(torch.from_numpy(cost).cuda(), torch.from_numpy(valid).cuda())
    torch.cuda.synchronize()
    return match.cpu().numpy(), status.cpu().numpy()


@pytest.mark.parametrize("G,Q", [(132, 256), (7, 24), (1, 1), (64, 64), (200, 1000), (3, 5)])
def test_match_equals_scipy_on_random_costs(G, Q):
    rng = np.random.default_rng(G * 1000 + Q)
    count = 12
    cost = rng.standard_normal((count, G, Q)).astype(np.float32) * 3
    valid = rng.random((count, G)) < rng.uniform(0.05, 0.9, (count, 1))
    valid[0] = True
    valid[1] = False
    valid[:, min(G, Q):] &= G <= Q            # never more targets than queries in these cases
    got, status = run_match(cost, valid)
    assert not status.any()
    for p in range(count):
        np.testing.assert_array_equal(got[p], scipy_match(cost[p], valid[p]), err_msg=f"problem {p}")


def test_match_equals_scipy_on_tie_heavy_costs():
    """Integer / constant / partly infinite matrices: ties everywhere, the scan order decides -- the
    kernel follows rectangular_lsap.cpp's rule, so even these come out identical."""
    rng = np.random.default_rng(5)
    count, G, Q = 24, 20, 40
    cost = rng.integers(0, 3, (count, G, Q)).astype(np.float32)
    cost[0] = 0.0
    cost[1] = 7.5
    cost[2:8][rng.random((6, G, Q)) < 0.3] = np.inf
    cost[8] = -0.0
    valid = rng.random((count, G)) < 0.7
    valid[0] = valid[1] = True
    got, status = run_match(cost, valid)
    for p in range(count):
        try:
            want = scipy_match(cost[p], valid[p])
        except ValueError:                      # infeasible for scipy -> status 1, all -1
            assert status[p] == 1 and (got[p] == -1).all()
            continue
        assert status[p] == 0
        np.testing.assert_array_equal(got[p], want, err_msg=f"problem {p}")


def test_match_rejects_what_scipy_rejects():
    cost = np.zeros((4, 3, 5), dtype=np.float32)
    cost[0, 1, 2] = np.nan
    cost[1, 0, 0] = -np.inf
    cost[2, 2, :] = np.inf                      # a target no query can take: infeasible
    valid = np.ones((4, 3), dtype=bool)
    got, status = run_match(cost, valid)
    assert status.tolist() == [1, 1, 1, 0]
    assert (got[:3] == -1).all() and sorted(got[3].tolist()) == [0, 1, 2]
    cost2 = np.zeros((1, 6, 4), dtype=np.float32)   # more valid targets than queries
    got, status = run_match(cost2, np.ones((1, 6), dtype=bool))
    assert status.tolist() == [1] and (got == -1).all()
    cost[0, 1, 2] = 0.0                         # a NaN in a row that is not valid does not matter
    cost[0, 0, 3] = np.nan
    valid[0, 0] = False
    got, status = run_match(cost, valid)
    assert status[0] == 0 and got[0, 0] == -1


@pytest.mark.parametrize("path", GOLD, ids=[p.split("losses_")[-1][:-4] for p in GOLD])
def test_criterion_on_gpu_reproduces_the_reference(path):
    from butd_detr_amd import losses as L
    z, ep, leaves, crit = load_case(path, device="cuda", grad=True)
    layers = int(z["meta_layers"])
    loss, out = L.compute_hungarian_loss(ep, layers, crit, int(z["meta_topk"]))
    np.testing.assert_array_equal(out["hungarian_match"].cpu().numpy(), z["out_match"])
    check_outputs(z, {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in out.items()
                      if k != "tokenized"}, L.hungarian_prefixes(layers), tol=5e-5)
    loss.backward()
    for name, t in leaves.items():
        want = z["grad_" + name]
        got = t.grad.cpu().numpy() if t.grad is not None else np.zeros_like(want)
        np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-6 + 1e-3 * np.abs(want).max(), err_msg=name)


def test_criterion_at_the_bench_size_reproduces_the_reference():
    """8 scenes x 256 queries x 7 prefixes, 1 .. 132 targets per scene, 256 token classes, 1024 seeds of 50 000 points
    (tests/golden/criterion_bench_size.npz, inputs regenerated from the reference run's seed): the device-side assignment
    of all 56 problems equals the reference's (scipy), every loss term and the gradients w.r.t. every prediction follow."""
    from butd_detr_amd import losses as L
    from tests.test_losses_cpu import bench_size_case, check_bench_size_grads
    z, ep, leaves, crit, shape = bench_size_case("cuda")
    loss, out = L.compute_hungarian_loss(ep, shape["layers"], crit, shape["topk"])
    np.testing.assert_array_equal(out["hungarian_match"].cpu().numpy(), z["out_match"])
    check_outputs(z, {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in out.items() if k != "tokenized"},
                  L.hungarian_prefixes(shape["layers"]), tol=5e-5)
    loss.backward()
    check_bench_size_grads(z, leaves, 1e-3)


def test_a_failed_assignment_makes_the_loss_nan():
    """Where scipy would raise (matcher.py:105: NaN costs) the device solver sets its status word; the criterion
    turns that into a NaN loss so that a graph replay cannot train on a silently wrong match."""
    from butd_detr_amd import losses as L
    z, ep, leaves, crit = load_case(GOLD[0], device="cuda", grad=False)
    layers = int(z["meta_layers"])
    loss, out = L.compute_hungarian_loss(dict(ep), layers, crit, int(z["meta_topk"]))
    assert torch.isfinite(loss) and not out["hungarian_status"].any()
    bad = dict(ep)
    key = "last_center" if "last_center" in bad else [k for k in bad if k.endswith("center")][0]
    bad[key] = bad[key].clone()
    bad[key][0, 0, 0] = float("nan")
    loss, out = L.compute_hungarian_loss(bad, layers, crit, int(z["meta_topk"]))
    assert out["hungarian_status"].any() and torch.isnan(loss)


def test_reference_style_api_returns_scipy_indices():
    """HungarianMatcher.forward / SetCriterion.forward with the reference's list-of-dict targets."""
    from scipy.optimize import linear_sum_assignment
    from butd_detr_amd import losses as L
    torch.manual_seed(0)
    B, Q, C = 3, 32, 40
    outputs = {"pred_logits": torch.randn(B, Q, C, device="cuda"),
               "pred_boxes": torch.cat([torch.rand(B, Q, 3, device="cuda") * 4, torch.rand(B, Q, 3, device="cuda") + 0.1], -1)}
    targets = []
    for n in (4, 0, 9):
        pm = torch.zeros(n, C, device="cuda")
        if n:
            pm[torch.arange(n), torch.randint(1, C - 1, (n,))] = 1.0
        targets.append({"labels": torch.randint(0, C, (n,), device="cuda"), "positive_map": pm,
                        "boxes": torch.cat([torch.rand(n, 3, device="cuda") * 4, torch.rand(n, 3, device="cuda") + 0.1], -1)})
    for soft in (True, False):
        matcher = L.HungarianMatcher(1, 5, 2, soft)
        indices = matcher(outputs, targets)
        tgt_boxes, pm, labels, valid = L._pad_targets(targets, C)
        cost = matcher.cost(outputs["pred_logits"], outputs["pred_boxes"], tgt_boxes, pm, labels).cpu().numpy()
        for b, (qi, ti) in enumerate(indices):
            n = len(targets[b]["boxes"])
            want_q, want_t = linear_sum_assignment(cost[b, :n].T)
            assert qi.dtype == torch.int64 and ti.dtype == torch.int64
            np.testing.assert_array_equal(qi.numpy(), want_q)
            np.testing.assert_array_equal(ti.numpy(), want_t)
    crit = L.SetCriterion(L.HungarianMatcher(1, 0, 2, True), losses=["boxes", "labels"])
    losses, indices = crit(outputs, targets)
    assert set(losses) == {"loss_ce", "loss_bbox", "loss_giou"} and len(indices) == B
    assert all(torch.isfinite(v) for v in losses.values())


def _north_star_case(seed=0, P=7, B=8, Q=256, C=256, Lt=80, D=64, N=20000):
    from butd_detr_amd.train_step import synthetic_ground_truth
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    pc = rng.uniform(-3, 3, (B, N, 3)).astype(np.float32)
    gt = {k: torch.from_numpy(v).cuda() for k, v in synthetic_ground_truth(pc, rng).items()}
    lens = torch.randint(Lt // 2, Lt + 1, (B,))
    out = {"pred_logits": torch.randn(P, B, Q, C, device="cuda") * 2,
           "pred_boxes": torch.cat([torch.rand(P, B, Q, 3, device="cuda") * 6 - 3,
                                    torch.rand(P, B, Q, 3, device="cuda") * 1.5 + 0.05], -1),
           "proj_queries": torch.nn.functional.normalize(torch.randn(P, B, Q, D, device="cuda"), dim=-1),
           "proj_tokens": torch.nn.functional.normalize(torch.randn(B, Lt, D, device="cuda"), dim=-1),
           "tokenized": {"attention_mask": (torch.arange(Lt)[None] < lens[:, None]).long().cuda()}}
    tgt = {"boxes": torch.cat([gt["center_label"], gt["size_gts"]], -1), "positive_map": gt["positive_map"],
           "labels": gt["sem_cls_label"] % C, "valid": gt["box_label_mask"] > 0}
    return out, tgt


@pytest.mark.parametrize("weights,soft", [((1, 0, 2), True), ((1, 5, 2), True), ((2, 5, 2), False)])
def test_fused_terms_equal_the_dense_torch_expressions(weights, soft):
    """The kernels of include/butd_criterion.h vs the dense torch expressions (which the CPU tests pin to the
    reference's vectors), at the bench shape: cost tensor, assignment, every loss term, every gradient."""
    from butd_detr_amd import losses as L
    out, tgt = _north_star_case()
    crit = L.SetCriterion(L.HungarianMatcher(*weights, soft), ["boxes", "labels", "contrastive_align"])
    res = {}
    for backend in ("torch", "hip"):
        L.set_backend(backend)
        try:
            leaves = {k: out[k].clone().requires_grad_(True) for k in ("pred_logits", "pred_boxes", "proj_queries",
                                                                       "proj_tokens")}
            o = dict(out, **leaves)
            cost = crit.matcher.cost(o["pred_logits"], o["pred_boxes"], tgt["boxes"], tgt["positive_map"],
                                     tgt["labels"], tgt["valid"])
            losses, match = crit.dense_forward(o, tgt)
            probe = torch.linspace(0.5, 1.5, 7, device="cuda")
            sum((v * probe).sum() for v in losses.values()).backward()
            res[backend] = (cost, match, {k: v.detach() for k, v in losses.items()},
                            {k: v.grad for k, v in leaves.items()})
        finally:
            L.set_backend("hip")
    (c0, m0, l0, g0), (c1, m1, l1, g1) = res["torch"], res["hip"]
    ok = tgt["valid"][None, :, :, None].expand_as(c0)
    assert torch.allclose(c1[ok], c0[ok], rtol=1e-5, atol=1e-5)
    assert (m0 == m1).float().mean() > 0.999          # a near-tie may flip under 1e-7 cost differences
    if not torch.equal(m0, m1):
        pytest.skip("assignment differs on a near-tie: loss comparison would not be like for like")
    for k in l0:
        assert torch.allclose(l1[k], l0[k], rtol=2e-4, atol=1e-5), k
    for k in g0:
        scale = g0[k].abs().max()
        assert (g1[k] - g0[k]).abs().max() <= 2e-4 * scale + 1e-7, k


def test_fused_seed_objectness_equals_the_dense_expression():
    from butd_detr_amd import losses as L
    torch.manual_seed(3)
    B, K, G, N, topk = 8, 1024, 132, 50000, 4
    n_valid = [1, 16, 5, 0, 30, 2, 9, 12]
    mask = torch.zeros(B, G, device="cuda")
    pil = torch.full((B, N), -1, dtype=torch.long, device="cuda")
    seed_inds = torch.stack([torch.randperm(N, device="cuda")[:K] for _ in range(B)]).int()
    for b, nv in enumerate(n_valid):
        mask[b, :nv] = 1
        owners = torch.randint(-1, max(nv, 1), (K,), device="cuda") if nv else torch.full((K,), -1, device="cuda")
        if nv:
            owners[: nv * topk] = torch.arange(nv, device="cuda").repeat_interleave(topk)   # >= topk members each
        pil[b, seed_inds[b].long()] = owners
    ep = {"box_label_mask": mask, "seed_inds": seed_inds, "seed_xyz": torch.rand(B, K, 3, device="cuda") * 6 - 3,
          "center_label": torch.rand(B, G, 3, device="cuda") * 6 - 3, "size_gts": torch.rand(B, G, 3, device="cuda") + 0.3,
          "point_instance_label": pil}
    base = torch.randn(B, 1, K, device="cuda")
    res = {}
    for backend in ("torch", "hip"):
        L.set_backend(backend)
        try:
            x = base.clone().requires_grad_(True)
            loss = L.compute_points_obj_cls_loss_hard_topk(dict(ep, seeds_obj_cls_logits=x), topk)
            (loss * 3.0).backward()
            res[backend] = (loss.detach(), x.grad)
        finally:
            L.set_backend("hip")
    assert torch.allclose(res["hip"][0], res["torch"][0], rtol=1e-5, atol=1e-7)
    assert (res["hip"][1] - res["torch"][1]).abs().max() <= 1e-5 * res["torch"][1].abs().max()


def test_single_node_tail_equals_the_term_by_term_graph():
    """compute_hungarian_loss at the bench shape: the one-node tail (butd_criterion_reduce / _scale) against the
    term-by-term autograd graph of the same kernels -- every end_points entry and every gradient, for an upstream
    gradient that is not 1, with and without the objectness term; and a failed assignment (NaN loss, zero gradients)."""
    from butd_detr_amd import losses as L
    out, tgt = _north_star_case()
    P, B, Q, C = out["pred_logits"].shape
    layers = P - 1
    prefixes = L.hungarian_prefixes(layers)
    torch.manual_seed(11)
    K, N, G = 1024, 50000, tgt["boxes"].shape[1]
    seed_inds = torch.stack([torch.randperm(N, device="cuda")[:K] for _ in range(B)]).int()
    pil = torch.randint(-1, 3, (B, N), device="cuda")
    base = {"box_label_mask": tgt["valid"].float(), "center_label": tgt["boxes"][..., :3].contiguous(),
            "size_gts": tgt["boxes"][..., 3:].contiguous(), "positive_map": tgt["positive_map"],
            "sem_cls_label": tgt["labels"], "proj_tokens": out["proj_tokens"], "tokenized": out["tokenized"],
            "seed_inds": seed_inds, "seed_xyz": torch.rand(B, K, 3, device="cuda") * 6 - 3, "point_instance_label": pil}
    crit = L.SetCriterion(L.HungarianMatcher(1, 5, 2, True), ["boxes", "labels", "contrastive_align"])
    for with_gen, poison in ((True, False), (False, False), (True, True)):
        res = {}
        for tail in (False, True):
            prev, L._TAIL[0] = L._TAIL[0], tail
            try:
                leaves = {"logits": out["pred_logits"].clone().requires_grad_(True),
                          "boxes": out["pred_boxes"].clone().requires_grad_(True),
                          "que": out["proj_queries"].clone().requires_grad_(True),
                          "seed": torch.randn(B, 1, K, device="cuda").requires_grad_(True) if with_gen else None}
                if tail is False:
                    seed0 = leaves["seed"]
                elif with_gen:
                    leaves["seed"] = seed0.detach().clone().requires_grad_(True)
                ep = dict(base)
                boxes = leaves["boxes"]
                if poison:
                    boxes = boxes.clone()
                    boxes.data[0, 0, 0, 0] = float("nan")
                    boxes = boxes.detach().requires_grad_(True)
                    leaves["boxes"] = boxes
                for i, pfx in enumerate(prefixes):
                    ep[f"{pfx}sem_cls_scores"] = leaves["logits"][i]
                    ep[f"{pfx}center"], ep[f"{pfx}pred_size"] = boxes[i, ..., :3], boxes[i, ..., 3:]
                    ep[f"{pfx}proj_queries"] = leaves["que"][i]
                if with_gen:
                    ep["seeds_obj_cls_logits"] = leaves["seed"]
                loss, ep = L.compute_hungarian_loss(ep, layers, crit, 4)
                (loss * 0.37).backward()
                keys = ["loss", "loss_ce", "loss_bbox", "loss_giou", "loss_constrastive_align", "query_points_generation_loss"]
                keys += [f"{pfx}_{k}" for pfx in prefixes for k in ("loss_ce", "loss_bbox", "loss_giou", "loss_contrastive_align")]
                res[tail] = ({k: ep[k].detach().clone() for k in keys}, ep["hungarian_match"].clone(),
                             {k: (None if v is None else v.grad) for k, v in leaves.items()})
            finally:
                L._TAIL[0] = prev
        (v0, m0, g0), (v1, m1, g1) = res[False], res[True]
        assert torch.equal(m0, m1)
        for k in v0:
            if poison and k == "loss":
                assert torch.isnan(v0[k]) and torch.isnan(v1[k])
            elif not poison:
                assert torch.allclose(v1[k], v0[k], rtol=2e-5, atol=1e-6), (k, float(v0[k]), float(v1[k]))
        for k in g0:
            if g0[k] is None:
                assert g1[k] is None
            elif poison:
                assert not g1[k].abs().sum() > 0, k            # nothing trains on an invalid match
            else:
                assert (g1[k] - g0[k]).abs().max() <= 2e-5 * g0[k].abs().max() + 1e-9, k
