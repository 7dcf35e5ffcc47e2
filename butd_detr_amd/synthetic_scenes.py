"""Seeded synthetic inputs shaped like the reference's ScanNet/SR3D batches.

Nothing here reads ScanNet: the reference loader (src/visual_data_handlers.py:84-126,
src/joint_det_dataset.py:405-453,738-790) fixes 50 000 points per scene (sampled WITH
replacement when a scan is smaller, hence exact duplicates), xyz jitter, mean-centred
colours, 132 detected-box slots and a tokenised utterance; these generators reproduce
those shapes and quirks so the index kernels see realistic ties and densities.
"""
import numpy as np

MAX_NUM_OBJ = 132  # src/joint_det_dataset.py:33


def _sample_box_surface(rng, n, lo, hi):
    """n points uniformly on the surface of the axis-aligned box [lo, hi]."""
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    ext = hi - lo
    areas = np.array([ext[1] * ext[2], ext[1] * ext[2], ext[0] * ext[2],
                      ext[0] * ext[2], ext[0] * ext[1], ext[0] * ext[1]])
    face = rng.choice(6, size=n, p=areas / areas.sum())
    pts = lo + rng.random((n, 3)) * ext
    axis = face // 2
    side = face % 2
    pts[np.arange(n), axis] = np.where(side == 0, lo[axis], hi[axis])
    return pts


def scannet_like_scene(seed=1184, n_points=50000, dup_frac=0.02, origin_point=True,
                       dense_cluster=False):
    """One (n_points, 6) float32 cloud: xyz on the surfaces of an 8x6x3 m room with 25
    furniture boxes, U(0, 5e-3) jitter, shuffled order, ``dup_frac`` exact duplicates and
    (optionally) one point at the origin; rgb in [-0.5, 0.5) minus its mean."""
    rng = np.random.default_rng(seed)
    n_unique = int(round(n_points * (1.0 - dup_frac)))
    room_lo, room_hi = np.array([-4.0, -3.0, 0.0]), np.array([4.0, 3.0, 3.0])
    n_room = int(n_unique * 0.55)
    n_furn = n_unique - n_room
    # room shell without the ceiling: floor + 4 walls
    shell = _sample_box_surface(rng, n_room * 2, room_lo, room_hi)
    shell = shell[shell[:, 2] < room_hi[2] - 1e-9][:n_room]
    while shell.shape[0] < n_room:
        extra = _sample_box_surface(rng, n_room, room_lo, room_hi)
        extra = extra[extra[:, 2] < room_hi[2] - 1e-9]
        shell = np.concatenate([shell, extra])[:n_room]
    parts = [shell]
    per = np.full(25, n_furn // 25)
    per[: n_furn - per.sum()] += 1
    for i in range(25):
        size = rng.uniform(0.4, 2.0, size=3)
        size[2] = min(size[2], 2.0)
        centre = np.array([rng.uniform(room_lo[0] + 1.0, room_hi[0] - 1.0),
                           rng.uniform(room_lo[1] + 1.0, room_hi[1] - 1.0),
                           size[2] / 2.0])
        if dense_cluster and i == 0:
            size = size * 0.25
            per[0] *= 10
        parts.append(_sample_box_surface(rng, int(per[i]), centre - size / 2, centre + size / 2))
    xyz = np.concatenate(parts)[:n_unique] if not dense_cluster else np.concatenate(parts)
    if xyz.shape[0] > n_unique:
        xyz = xyz[rng.permutation(xyz.shape[0])[:n_unique]]
    xyz = xyz + rng.uniform(0.0, 5e-3, size=xyz.shape)  # joint_det_dataset.py:386-388
    xyz = xyz.astype(np.float32)
    if origin_point:
        xyz[0] = 0.0
    n_dup = n_points - xyz.shape[0]
    if n_dup > 0:  # visual_data_handlers.py:113-118: sampling with replacement
        xyz = np.concatenate([xyz, xyz[rng.integers(0, xyz.shape[0], size=n_dup)]])
    perm = rng.permutation(n_points)
    xyz = xyz[perm]
    rgb = rng.random((n_points, 3)).astype(np.float32) - 0.5
    rgb = rgb - rgb.mean(0, keepdims=True)  # joint_det_dataset.py:68,414-415
    return np.concatenate([xyz, rgb.astype(np.float32)], axis=1).astype(np.float32)


def scene_batch(batch, seed=1184, n_points=50000, **kw):
    """(batch, n_points, 6) float32; scene i uses seed+i."""
    return np.stack([scannet_like_scene(seed + i, n_points, **kw) for i in range(batch)])


def uniform_cloud(seed=0, n_points=4096, batch=1):
    """Config-1 plumbing cloud: xyz ~ U(-2,2)^3, rgb ~ U(-0.5,0.5)."""
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-2.0, 2.0, size=(batch, n_points, 3))
    rgb = rng.uniform(-0.5, 0.5, size=(batch, n_points, 3))
    return np.concatenate([xyz, rgb], axis=-1).astype(np.float32)


def detected_boxes(batch, seed=0, min_valid=10, max_valid=60, num_class=485):
    """det_boxes (B,132,6) f32, det_bbox_label_mask (B,132) bool, det_class_ids (B,132) i64
    (the keys train_dist_mod.py:103-110 feeds the model)."""
    rng = np.random.default_rng(seed + 7919)
    boxes = np.zeros((batch, MAX_NUM_OBJ, 6), dtype=np.float32)
    mask = np.zeros((batch, MAX_NUM_OBJ), dtype=bool)
    cls = np.zeros((batch, MAX_NUM_OBJ), dtype=np.int64)
    for b in range(batch):
        nv = int(rng.integers(min_valid, max_valid + 1))
        boxes[b, :nv, 0] = rng.uniform(-3.5, 3.5, nv)
        boxes[b, :nv, 1] = rng.uniform(-2.5, 2.5, nv)
        boxes[b, :nv, 2] = rng.uniform(0.1, 2.0, nv)
        boxes[b, :nv, 3:] = rng.uniform(0.2, 2.0, (nv, 3))
        mask[b, :nv] = True
        cls[b, :nv] = rng.integers(0, num_class, nv)
    return boxes, mask, cls


def token_batch(batch, seed=0, length=80, vocab=50265, max_pad_frac=0.3):
    """input_ids (B,L) i64 and attention_mask (B,L) i64 with 0-30 % right padding; at least
    one row is unpadded (``padding='longest'`` semantics, bdetr.py:164-166)."""
    rng = np.random.default_rng(seed + 104729)
    ids = rng.integers(3, vocab, size=(batch, length)).astype(np.int64)
    att = np.ones((batch, length), dtype=np.int64)
    for b in range(1, batch):
        npad = int(rng.integers(0, int(length * max_pad_frac) + 1))
        if npad:
            ids[b, length - npad:] = 1  # RoBERTa <pad>
            att[b, length - npad:] = 0
    ids[:, 0] = 0   # <s>
    for b in range(batch):
        last = int(att[b].sum()) - 1
        ids[b, last] = 2  # </s>
    return ids, att
