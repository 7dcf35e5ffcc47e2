"""CPU (-m "not gpu"): host logic of the criterion (butd_detr_amd/losses.py) against vectors captured from the
reference's models/losses.py (tests/golden/make_losses_golden.py), and the assignment oracle against scipy.

The product's assignment runs only in butd_hungarian_match (HIP); here the golden assignment is injected,
and separately the product's cost tensor + the CPU oracle must reproduce it."""
import glob
import os

import numpy as np
import pytest
import torch

from butd_detr_amd import losses as L
from oracle import lsap_oracle

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "losses_*.npz")))


def load_case(path, device="cpu", grad=False):
    z = np.load(path)
    ep, leaves = {}, {}
    for k in z.files:
        if not k.startswith("in_"):
            continue
        t = torch.from_numpy(z[k]).to(device)
        name = k[3:]
        if name == "attention_mask":
            ep["tokenized"] = {"attention_mask": t}
            continue
        if grad and ("grad_" + name) in z.files:
            t = t.clone().requires_grad_(True)
            leaves[name] = t
        ep[name] = t
    weights = tuple(float(w) for w in z["meta_weights"])
    names = ["boxes", "labels"] + (["contrastive_align"] if int(z["meta_contrastive"]) else [])
    crit = L.SetCriterion(L.HungarianMatcher(*weights, True), losses=names, eos_coef=0.1, temperature=0.07)
    return z, ep, leaves, crit


def check_outputs(z, out, prefixes, tol=2e-5):
    for k in ("loss", "loss_ce", "loss_bbox", "loss_giou", "query_points_generation_loss",
              "loss_constrastive_align"):
        np.testing.assert_allclose(float(out[k]), float(z["out_" + k]), rtol=tol, atol=tol, err_msg=k)
    for p in prefixes:
        for key in ("loss_ce", "loss_bbox", "loss_giou", "loss_contrastive_align"):
            if f"out_{p}_{key}" in z.files:
                np.testing.assert_allclose(float(out[f"{p}_{key}"]), float(z[f"out_{p}_{key}"]), rtol=tol,
                                           atol=tol, err_msg=p + key)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[7:-4] for p in GOLD])
def test_dense_criterion_reproduces_the_reference_given_its_assignment(path):
    z, ep, leaves, crit = load_case(path, grad=True)
    layers = int(z["meta_layers"])
    loss, out = L.compute_hungarian_loss(ep, layers, crit, int(z["meta_topk"]),
                                         match=torch.from_numpy(z["out_match"]))
    check_outputs(z, out, L.hungarian_prefixes(layers))
    loss.backward()
    for name, t in leaves.items():
        want = z["grad_" + name]
        got = t.grad.numpy() if t.grad is not None else np.zeros_like(want)
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-6 + 1e-4 * np.abs(want).max(), err_msg=name)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[7:-4] for p in GOLD])
def test_cost_tensor_and_oracle_reproduce_the_reference_assignment(path):
    z, ep, _, crit = load_case(path)
    prefixes = L.hungarian_prefixes(int(z["meta_layers"]))
    logits = torch.stack([ep[f"{p}sem_cls_scores"] for p in prefixes])
    boxes = torch.stack([torch.cat([ep[f"{p}center"], ep[f"{p}pred_size"]], -1) for p in prefixes])
    tgt = torch.cat([ep["center_label"], ep["size_gts"]], -1)
    cost = crit.matcher.cost(logits, boxes, tgt, ep["positive_map"], ep["sem_cls_label"]).numpy()
    valid = ep["box_label_mask"].numpy() > 0
    assert cost.shape == (len(prefixes),) + valid.shape + (logits.shape[2],)
    for p in range(cost.shape[0]):
        for b in range(cost.shape[1]):
            np.testing.assert_array_equal(lsap_oracle.match_targets(cost[p, b], valid[b]), z["out_match"][p, b])


def test_oracle_is_scipy_bit_for_bit_including_ties():
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(0)
    for t in range(200):
        nr, nc = int(rng.integers(1, 20)), int(rng.integers(1, 20))
        c = [rng.standard_normal((nr, nc)), rng.integers(0, 3, (nr, nc)), np.zeros((nr, nc)),
             np.where(rng.random((nr, nc)) < 0.2, np.inf, rng.integers(0, 5, (nr, nc)))][t % 4].astype(np.float32)
        try:
            want = linear_sum_assignment(c)
        except ValueError:
            with pytest.raises(ValueError):
                lsap_oracle.solve(c)
            continue
        got = lsap_oracle.solve(c)
        np.testing.assert_array_equal(got[0], want[0])
        np.testing.assert_array_equal(got[1], want[1])
    for bad in (np.array([[np.nan, 1.0]]), np.array([[-np.inf, 1.0]])):
        with pytest.raises(ValueError):
            lsap_oracle.solve(bad)


def test_matcher_has_no_cpu_fallback():
    with pytest.raises(RuntimeError, match="CPU not supported"):
        L.hungarian_match(torch.zeros(1, 2, 4), torch.ones(1, 2, dtype=torch.bool))


def test_box_helpers():
    a = torch.tensor([[0.0, 0, 0, 2, 2, 2], [5.0, 5, 5, 1, 1, 1]])
    corners = L.box_cxcyczwhd_to_xyzxyz(a)
    assert torch.equal(corners[0], torch.tensor([-1.0, -1, -1, 1, 1, 1]))
    g = L.generalized_box_iou3d(corners, corners)
    assert torch.allclose(torch.diag(g), torch.ones(2))
    assert g[0, 1] < 0      # disjoint boxes: negative GIoU


def bench_size_case(device="cpu"):
    """The bench-size criterion case (tests/golden/criterion_bench_size.npz): its inputs are regenerated from the seed the
    reference run used (make_losses_golden.BENCH_SHAPE), only the reference's outputs are stored."""
    from tests.golden.make_losses_golden import BENCH_SHAPE, make_inputs
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "criterion_bench_size.npz"))
    ep, _ = make_inputs(**BENCH_SHAPE)
    leaves = {}
    for k, v in list(ep.items()):
        if k == "tokenized":
            ep[k] = {"attention_mask": v["attention_mask"].to(device)}
            continue
        t = v.to(device)
        if ("grad8_" + k) in z.files:
            t = t.clone().requires_grad_(True)
            leaves[k] = t
        ep[k] = t
    crit = L.SetCriterion(L.HungarianMatcher(1, 0, 2, True), losses=["boxes", "labels", "contrastive_align"], eos_coef=0.1,
                          temperature=0.07)
    return z, ep, leaves, crit, BENCH_SHAPE


def check_bench_size_grads(z, leaves, rtol):
    for name, t in leaves.items():
        g = t.grad if t.grad is not None else torch.zeros_like(t)
        want = z["grad8_" + name]
        got = (g[:, ::8] if g.dim() == 3 and g.shape[1] >= 64 else g).detach().cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=rtol, atol=1e-7 + rtol * np.abs(want).max(), err_msg=name)
        total = float(g.detach().double().abs().sum())
        np.testing.assert_allclose(total, float(z["gradsum_" + name]), rtol=10 * rtol, err_msg=name + " (sum of |grad|)")


def test_dense_criterion_at_the_bench_size_given_the_reference_assignment():
    z, ep, leaves, crit, shape = bench_size_case()
    loss, out = L.compute_hungarian_loss(ep, shape["layers"], crit, shape["topk"], match=torch.from_numpy(z["out_match"]))
    check_outputs(z, out, L.hungarian_prefixes(shape["layers"]), tol=5e-5)
    loss.backward()
    check_bench_size_grads(z, leaves, 1e-4)
