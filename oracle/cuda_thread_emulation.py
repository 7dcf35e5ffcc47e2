"""Thread-by-thread emulation of the reference CUDA kernels -- TEST INFRASTRUCTURE ONLY.

A second, independent restatement used to pin ``oracle/pointnet2_oracle.c``: instead of
re-deriving what a block computes, it *executes* the CUDA source structure literally in
Python -- one Python loop iteration per CUDA thread, explicit ``__shared__`` arrays, the
``__syncthreads()``-separated tree levels in program order -- with numpy float32 scalars so
every operation rounds once to fp32 like the (uncontracted) device code.  Pure-Python loops:
use small sizes only (n up to a few thousand).

Line references: /root/reference/pointnet2/_ext_src/src/{sampling_gpu,ball_query_gpu,interpolate_gpu}.cu
"""
import math

import numpy as np

F = np.float32
TOTAL_THREADS = 512


def opt_n_threads(work_size):
    """include/cuda_utils.h:18-24."""
    pow_2 = int(math.log(float(work_size)) / math.log(2.0))
    return max(min(1 << pow_2, TOTAL_THREADS), 1)


def furthest_point_sampling(dataset, m):
    """sampling_gpu.cu:74-178 with block_size = opt_n_threads(n) (sampling_gpu.cu:183)."""
    dataset = np.asarray(dataset, dtype=np.float32)
    b, n, _ = dataset.shape
    block_size = opt_n_threads(n)
    idxs = np.zeros((b, m), dtype=np.int32)
    for batch_index in range(b):
        pts = dataset[batch_index]
        temp = np.full(n, 1e10, dtype=np.float32)
        if m <= 0:
            continue
        dists = np.zeros(block_size, dtype=np.float32)    # __shared__ float dists[block_size]
        dists_i = np.zeros(block_size, dtype=np.int32)     # __shared__ int dists_i[block_size]
        old = 0
        idxs[batch_index, 0] = old
        for j in range(1, m):
            x1, y1, z1 = pts[old, 0], pts[old, 1], pts[old, 2]
            for tid in range(block_size):                   # every CUDA thread
                besti = 0
                best = F(-1)
                for k in range(tid, n, block_size):
                    x2, y2, z2 = pts[k, 0], pts[k, 1], pts[k, 2]
                    mag = F(F(F(x2 * x2) + F(y2 * y2)) + F(z2 * z2))
                    if float(mag) <= 1e-3:                  # double compare, :106
                        continue
                    dx, dy, dz = F(x2 - x1), F(y2 - y1), F(z2 - z1)
                    d = F(F(F(dx * dx) + F(dy * dy)) + F(dz * dz))
                    d2 = min(d, temp[k])
                    temp[k] = d2
                    besti = k if d2 > best else besti
                    best = d2 if d2 > best else best
                dists[tid] = best
                dists_i[tid] = besti
            # __syncthreads(); tree, :119-173
            s = block_size // 2
            while s >= 1:
                for tid in range(s):                        # if (tid < s) __update(tid, tid + s)
                    v1, v2 = dists[tid], dists[tid + s]
                    i1, i2 = dists_i[tid], dists_i[tid + s]
                    dists[tid] = max(v1, v2)
                    dists_i[tid] = i2 if v2 > v1 else i1
                s //= 2
            old = int(dists_i[0])
            idxs[batch_index, j] = old
    return idxs


def ball_query(new_xyz, xyz, radius, nsample):
    """ball_query_gpu.cu:14-49."""
    new_xyz = np.asarray(new_xyz, dtype=np.float32)
    xyz = np.asarray(xyz, dtype=np.float32)
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    idx = np.zeros((b, m, nsample), dtype=np.int32)
    radius2 = F(F(radius) * F(radius))
    for bi in range(b):
        for j in range(m):
            nx, ny, nz = new_xyz[bi, j]
            cnt = 0
            k = 0
            while k < n and cnt < nsample:
                x, y, z = xyz[bi, k]
                dx, dy, dz = F(nx - x), F(ny - y), F(nz - z)
                d2 = F(F(F(dx * dx) + F(dy * dy)) + F(dz * dz))
                if d2 < radius2:
                    if cnt == 0:
                        idx[bi, j, :] = k
                    idx[bi, j, cnt] = k
                    cnt += 1
                k += 1
    return idx


def three_nn(unknown, known):
    """interpolate_gpu.cu:14-64 (double best*, float d)."""
    unknown = np.asarray(unknown, dtype=np.float32)
    known = np.asarray(known, dtype=np.float32)
    b, n, _ = unknown.shape
    m = known.shape[1]
    dist2 = np.zeros((b, n, 3), dtype=np.float32)
    idx = np.zeros((b, n, 3), dtype=np.int32)
    for bi in range(b):
        for j in range(n):
            ux, uy, uz = unknown[bi, j]
            best1 = best2 = best3 = 1e40
            b1 = b2 = b3 = 0
            for k in range(m):
                x, y, z = known[bi, k]
                dx, dy, dz = F(ux - x), F(uy - y), F(uz - z)
                d = float(F(F(F(dx * dx) + F(dy * dy)) + F(dz * dz)))
                if d < best1:
                    best3, b3 = best2, b2
                    best2, b2 = best1, b1
                    best1, b1 = d, k
                elif d < best2:
                    best3, b3 = best2, b2
                    best2, b2 = d, k
                elif d < best3:
                    best3, b3 = d, k
            with np.errstate(over="ignore"):
                dist2[bi, j] = (np.float32(best1), np.float32(best2), np.float32(best3))
            idx[bi, j] = (b1, b2, b3)
    return dist2, idx
