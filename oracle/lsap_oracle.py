"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the assignment step of the criterion.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
path (butd_detr_amd/losses.py -> butd_hungarian_match) never does.

What it restates.  HungarianMatcher.forward (/root/reference/models/losses.py:306-320) hands the cost
matrix to scipy.optimize.linear_sum_assignment -- a THIRD-PARTY dependency (scipy 1.7.3 pinned in
environment.yml:92; 1.15 in this image), not code of the reference.  Its published algorithm is Crouse's
rectangular shortest-augmenting-path variant of Jonker-Volgenant (scipy/optimize/rectangular_lsap/
rectangular_lsap.cpp); `solve` below restates it in pure Python (small cases only).

Pinning.  scipy itself is importable here and on the GPU box, so the restatement and the HIP kernel are both
checked against scipy's own answers (tests/test_losses_cpu.py, tests/test_gpu_losses.py), and the complete
criterion against outputs of the reference's losses.py captured by tests/golden/make_losses_golden.py.
"""
import math

import numpy as np


def solve(cost):
    """cost (nr, nc) array-like -> (row_ind, col_ind) exactly as scipy.optimize.linear_sum_assignment
    (minimisation).  Raises ValueError like scipy on NaN / -inf entries or an infeasible matrix."""
    cost = np.asarray(cost, dtype=np.float64)
    nr, nc = cost.shape
    if nr == 0 or nc == 0:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    transpose = nc < nr
    if transpose:                                   # rectangular_lsap.cpp: work on the wide orientation
        cost = np.ascontiguousarray(cost.T)
        nr, nc = nc, nr
    if np.isnan(cost).any() or (cost == -np.inf).any():
        raise ValueError("matrix contains invalid numeric entries")
    u = [0.0] * nr
    v = [0.0] * nc
    shortest = [math.inf] * nc
    path = [-1] * nc
    col4row = [-1] * nr
    row4col = [-1] * nc
    for cur in range(nr):
        # ---- augmenting_path(cur)
        min_val = 0.0
        remaining = [nc - it - 1 for it in range(nc)]     # reverse fill (constant matrix -> identity)
        num_remaining = nc
        SR = [False] * nr
        SC = [False] * nc
        shortest = [math.inf] * nc
        sink, i = -1, cur
        while sink == -1:
            index, lowest = -1, math.inf
            SR[i] = True
            for it in range(num_remaining):
                j = remaining[it]
                r = min_val + cost[i, j] - u[i] - v[j]
                if r < shortest[j]:
                    path[j] = i
                    shortest[j] = r
                # an equal value wins when its column is still free (new sink)
                if shortest[j] < lowest or (shortest[j] == lowest and row4col[j] == -1):
                    lowest = shortest[j]
                    index = it
            min_val = lowest
            if min_val == math.inf:
                raise ValueError("cost matrix is infeasible")
            j = remaining[index]
            if row4col[j] == -1:
                sink = j
            else:
                i = row4col[j]
            SC[j] = True
            num_remaining -= 1
            remaining[index] = remaining[num_remaining]
        # ---- dual update
        u[cur] += min_val
        for i in range(nr):
            if SR[i] and i != cur:
                u[i] += min_val - shortest[col4row[i]]
        for j in range(nc):
            if SC[j]:
                v[j] -= min_val - shortest[j]
        # ---- augment
        j = sink
        while True:
            i = path[j]
            row4col[j] = i
            col4row[i], j = j, col4row[i]
            if i == cur:
                break
    rows = np.arange(nr, dtype=np.int64)
    cols = np.asarray(col4row, dtype=np.int64)
    if transpose:
        order = np.argsort(cols)
        return cols[order], rows[order]
    return rows, cols


def match_targets(cost_gq, valid):
    """The dense interface of butd_hungarian_match for ONE problem: cost_gq (ng, nq) (targets x queries),
    valid (ng,) bool -> match (ng,) int32, query of every valid slot, -1 elsewhere."""
    cost_gq = np.asarray(cost_gq)
    valid = np.asarray(valid).astype(bool)
    match = np.full(valid.shape[0], -1, dtype=np.int32)
    slots = np.nonzero(valid)[0]
    if slots.size == 0:
        return match
    # the reference solves the (queries x compacted targets) matrix: C[b].T of this one
    rows, cols = solve(np.ascontiguousarray(cost_gq[slots].T))   # rows = queries, cols = compacted targets
    match[slots[cols]] = rows.astype(np.int32)
    return match


def scipy_match_dense(matcher):
    """bench.py cpu_baseline leg / tests only: a drop-in for HungarianMatcher.match_dense that does what the
    reference does -- cost to the host, scipy.optimize.linear_sum_assignment per (prefix, scene)."""
    import torch
    from scipy.optimize import linear_sum_assignment

    def match_dense(pred_logits, pred_boxes, tgt_boxes, positive_map, valid, labels=None):
        cost = matcher.cost(pred_logits, pred_boxes, tgt_boxes, positive_map, labels).detach().cpu().numpy()
        ok = valid.cpu().numpy().astype(bool)
        lead = cost.shape[:-2]
        flat = cost.reshape((-1,) + cost.shape[-2:])
        okf = np.broadcast_to(ok, lead + ok.shape[-1:]).reshape(-1, ok.shape[-1])
        match = -np.ones(okf.shape, dtype=np.int32)
        for p in range(flat.shape[0]):
            slots = np.nonzero(okf[p])[0]
            if slots.size:
                q, t = linear_sum_assignment(flat[p][slots].T)
                match[p, slots[t]] = q
        return torch.from_numpy(match.reshape(lead + ok.shape[-1:])).to(pred_logits.device)
    return match_dense
