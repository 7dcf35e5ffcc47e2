"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's per-sample augmentation.

Only tests/ (and the smoke / cpu_baseline legs) may import this; the product path is
butd_detr_amd/device_augment.py -> include/butd_augment.h.

Follows /root/reference/src/joint_det_dataset.py: `_augment` (:358-403), rot_x / rot_y / rot_z (:930-966),
box2points / points2box (:969-990), the detected-box transform (:595-607), `_get_target_boxes` (:497-522) with
visual_data_handlers.py:245-258, keeping numpy's dtypes: clouds are float32 arrays updated in place with
float64 operands (one rounding to float32 per step), boxes stay float64.
Pinned by tests/golden/augment_*.npz, captured from the reference itself (tests/golden/make_augment_golden.py).
"""
import numpy as np

MEAN_RGB = np.array([109.8, 97.2, 83.8]) / 256          # joint_det_dataset.py:68


def rot_matrix(axis, theta_deg):
    t = theta_deg * np.pi / 180
    c, s = np.cos(t), np.sin(t)
    if axis == "x":
        return np.array([[1.0, 0, 0], [0, c, -s], [0, s, c]])
    if axis == "y":
        return np.array([[c, 0, s], [0, 1.0, 0], [-s, 0, c]])
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


def draw(rotate, n_points, with_color=True, rng=np.random):
    """The random numbers of `_augment` in the order it draws them."""
    a = {"yz_flip": False, "xz_flip": False}
    if rotate:
        a["theta_z"] = 90 * rng.randint(0, 4) + 10 * rng.rand() - 5
        a["yz_flip"] = bool(rng.random() > 0.5)
        a["xz_flip"] = bool(rng.random() > 0.5)
    else:
        a["theta_z"] = (2 * rng.rand() - 1) * 5
    a["theta_x"] = (2 * rng.rand() - 1) * 2.5
    a["theta_y"] = (2 * rng.rand() - 1) * 2.5
    a["noise"] = rng.rand(n_points, 3) * 5e-3
    a["shift"] = rng.random((3,))[None, :] - 0.5
    a["scale"] = 0.98 + 0.04 * rng.random()
    if with_color:
        a["color_gain"] = 0.98 + 0.04 * rng.random((n_points, 3))
    return a


def augment_points(pc, color, a):
    """pc (N,3) float32, color (N,3) float32 or None (mean-subtracted) -> augmented copies."""
    pc = pc.copy()
    if a["yz_flip"]:
        pc[:, 0] = -pc[:, 0]
    if a["xz_flip"]:
        pc[:, 1] = -pc[:, 1]
    for axis in ("z", "x", "y"):
        pc[:, :3] = np.matmul(rot_matrix(axis, a["theta_" + axis]), pc[:, :3].T).T
    pc[:, :3] = pc[:, :3] + a["noise"]
    pc[:, :3] += a["shift"]
    pc[:, :3] *= a["scale"]
    if color is not None:
        color = color.copy()
        color += MEAN_RGB
        color *= a["color_gain"]
        color -= MEAN_RGB
    return pc, color


def augment_boxes(boxes, a):
    """boxes (D,6) centre+size -> transformed axis-aligned hulls (float64), :595-607."""
    boxes = np.asarray(boxes, dtype=np.float64)
    lo, hi = boxes[:, :3] - boxes[:, 3:] / 2, boxes[:, :3] + boxes[:, 3:] / 2
    corners = np.stack([np.stack([(hi if k & 2 else lo)[:, 0], (hi if k & 1 else lo)[:, 1],
                                  (hi if k & 4 else lo)[:, 2]], 1) for k in range(8)], 1)   # (D,8,3)
    pts = corners.reshape(-1, 3)
    for axis in ("z", "x", "y"):
        pts = np.matmul(rot_matrix(axis, a["theta_" + axis]), pts.T).T
    if a["yz_flip"]:
        pts[:, 0] = -pts[:, 0]
    if a["xz_flip"]:
        pts[:, 1] = -pts[:, 1]
    pts = pts + a["shift"]
    pts = pts * a["scale"]
    pts = pts.reshape(-1, 8, 3)
    return np.concatenate(((pts.min(1) + pts.max(1)) / 2, pts.max(1) - pts.min(1)), axis=1)


def instance_boxes(pc, instance, slots, jitter=None):
    """Target boxes of instance ids 0..slots-1 (:497-522): hull of each id's points (float32 arithmetic of
    visual_data_handlers.py:245-258), centre+size in float64, optional jitter, padding centre 1000."""
    boxes = np.zeros((slots, 6))
    mask = np.zeros(slots)
    for t in range(slots):
        sel = pc[instance == t, :3]
        if len(sel) == 0:
            continue
        mx, mn = np.max(sel, axis=0), np.min(sel, axis=0)
        ctr = (mx + mn) / 2.0
        length = mx - mn
        corners = np.concatenate([ctr - length / 2.0, ctr + length / 2.0])     # float32 if pc is
        boxes[t] = corners
        mask[t] = 1
    boxes = np.concatenate(((boxes[:, :3] + boxes[:, 3:]) * 0.5, boxes[:, 3:] - boxes[:, :3]), 1)
    if jitter is not None:
        boxes[mask > 0] *= jitter[mask > 0]
    boxes[mask == 0, :3] = 1000
    return boxes, mask
