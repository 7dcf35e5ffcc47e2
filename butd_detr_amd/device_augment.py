"""Scene augmentation on clouds that stay in HBM.

The reference augments on DataLoader workers in numpy (src/joint_det_dataset.py:358-403, 595-607, 497-522):
flip / rotate / jitter / shift / scale 50 000 points per sample, transform the 132 detected boxes the same way,
recompute the target boxes from the augmented object points -- then copies the cloud to the device.  Here the
host draws only the per-scene parameters (`draw_scene`, the same distributions in the same order, so a seeded
``numpy.random`` reproduces the reference sample for sample) and three launches of include/butd_augment.h do
the arithmetic on resident tensors.  No CPU fallback: the kernels are the product.
"""
import ctypes

import numpy as np
import torch

from . import _hiplib

MEAN_RGB = (109.8 / 256, 97.2 / 256, 83.8 / 256)      # joint_det_dataset.py:68


def _rot(axis, theta_deg):
    """rot_x / rot_y / rot_z (joint_det_dataset.py:930-966) as row-major 3x3 float64."""
    t = theta_deg * np.pi / 180
    c, s = float(np.cos(t)), float(np.sin(t))
    if axis == "x":
        return [1.0, 0.0, 0.0, 0.0, c, -s, 0.0, s, c]
    if axis == "y":
        return [c, 0.0, s, 0.0, 1.0, 0.0, -s, 0.0, c]
    return [c, -s, 0.0, s, c, 0.0, 0.0, 0.0, 1.0]


def draw_scene(rotate, rng=np.random):
    """The scalar draws of `_augment` (:362-396) in its order, WITHOUT the per-point noise / colour gains
    (the kernel draws those from a counter hash; pass arrays to `augment` to impose numpy's)."""
    a = {"yz_flip": False, "xz_flip": False}
    if rotate:
        a["theta_z"] = 90 * rng.randint(0, 4) + 10 * rng.rand() - 5
        a["yz_flip"] = bool(rng.random() > 0.5)
        a["xz_flip"] = bool(rng.random() > 0.5)
    else:
        a["theta_z"] = (2 * rng.rand() - 1) * 5
    a["theta_x"] = (2 * rng.rand() - 1) * 2.5
    a["theta_y"] = (2 * rng.rand() - 1) * 2.5
    a["shift"] = rng.random((3,)) - 0.5
    a["scale"] = 0.98 + 0.04 * rng.random()
    return a


def pack_params(scenes, device):
    """list of per-scene dicts (theta_z/x/y, yz_flip, xz_flip, shift (3,), scale) -> device buffer of
    ``butd_scene_augment``."""
    arr = (_hiplib.SceneAugment * len(scenes))()
    for s, a in zip(arr, scenes):
        s.rz[:], s.rx[:], s.ry[:] = _rot("z", a["theta_z"]), _rot("x", a["theta_x"]), _rot("y", a["theta_y"])
        s.shift[:] = [float(v) for v in np.asarray(a["shift"]).reshape(3)]
        s.scale = float(a["scale"])
        s.flip_yz, s.flip_xz = int(bool(a["yz_flip"])), int(bool(a["xz_flip"]))
    raw = np.frombuffer(ctypes.string_at(ctypes.addressof(arr), ctypes.sizeof(arr)), dtype=np.uint8).copy()
    return torch.from_numpy(raw).to(device)


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: CPU not supported (include/butd_augment.h runs on the device)")


def augment_points(point_clouds, params, has_color=True, noise=None, color_gain=None, seed=0, out=None):
    """point_clouds (B, N, 3+C) fp32 CUDA, params from `pack_params` -> augmented cloud (same shape).
    ``noise`` / ``color_gain`` (B, N, 3) float64 impose the reference's per-point draws; None = drawn on the
    device from ``seed``."""
    _need_gpu(point_clouds, "augment_points")
    pc = point_clouds.contiguous().float()
    B, N, ld = pc.shape
    out = torch.empty_like(pc) if out is None else out
    nz = None if noise is None else noise.to(pc.device, torch.float64).contiguous()
    cg = None if color_gain is None else color_gain.to(pc.device, torch.float64).contiguous()
    lib = _hiplib.load()
    with torch.cuda.device(pc.device):
        err = lib.butd_augment_points(B, N, ld - 3, int(bool(has_color and ld >= 6)), pc.data_ptr(),
                                      params.data_ptr(), nz.data_ptr() if nz is not None else None,
                                      cg.data_ptr() if cg is not None else None, *MEAN_RGB, int(seed),
                                      out.data_ptr(), _stream(pc))
    _hiplib.check(err, "butd_augment_points")
    return out


def augment_boxes(boxes, params):
    """Detected boxes (B, D, 6) centre+size fp32 CUDA -> transformed hulls (joint_det_dataset.py:595-607)."""
    _need_gpu(boxes, "augment_boxes")
    bx = boxes.contiguous().float()
    out = torch.empty_like(bx)
    lib = _hiplib.load()
    with torch.cuda.device(bx.device):
        err = lib.butd_augment_boxes(bx.shape[0], bx.shape[1], bx.data_ptr(), params.data_ptr(), out.data_ptr(),
                                     _stream(bx))
    _hiplib.check(err, "butd_augment_boxes")
    return out


def instance_boxes(point_clouds, point_instance_label, slots=132, jitter=None):
    """Target boxes from the (augmented) cloud: (center_label+size (B, slots, 6), box_label_mask (B, slots))
    as `_get_target_boxes` (:497-522) derives them; ``jitter`` (B, slots, 6) float64 = 0.95 + 0.1 * random."""
    _need_gpu(point_clouds, "instance_boxes")
    pc = point_clouds.contiguous().float()
    B, N, ld = pc.shape
    inst = point_instance_label.to(torch.int64).contiguous()
    jit = None if jitter is None else jitter.to(pc.device, torch.float64).contiguous()
    scratch = torch.empty((B, slots, 6), dtype=torch.int32, device=pc.device)
    boxes = torch.empty((B, slots, 6), device=pc.device)
    mask = torch.empty((B, slots), device=pc.device)
    lib = _hiplib.load()
    with torch.cuda.device(pc.device):
        err = lib.butd_instance_boxes(B, N, ld, slots, pc.data_ptr(), inst.data_ptr(),
                                      jit.data_ptr() if jit is not None else None, scratch.data_ptr(),
                                      boxes.data_ptr(), mask.data_ptr(), _stream(pc))
    _hiplib.check(err, "butd_instance_boxes")
    return boxes, mask


def augment_batch(inputs, targets, rotate, rng=np.random, seed=0):
    """One training batch, resident on the device: draws per-scene parameters on the host and rewrites
    ``point_clouds``, ``det_boxes`` and the target boxes (``center_label`` / ``size_gts`` / ``box_label_mask``
    from ``point_instance_label``) as the reference's dataset would have produced them."""
    pc = inputs["point_clouds"]
    B = pc.shape[0]
    scenes = [draw_scene(bool(rotate[b]) if hasattr(rotate, "__len__") else bool(rotate), rng) for b in range(B)]
    params = pack_params(scenes, pc.device)
    out_inputs, out_targets = dict(inputs), dict(targets)
    out_inputs["point_clouds"] = augment_points(pc, params, seed=seed)
    if "det_boxes" in inputs:
        out_inputs["det_boxes"] = augment_boxes(inputs["det_boxes"], params)
    if "point_instance_label" in targets:
        slots = targets["center_label"].shape[1] if "center_label" in targets else 132
        jitter = torch.from_numpy(0.95 + 0.1 * rng.random((B, slots, 6)))
        boxes, mask = instance_boxes(out_inputs["point_clouds"], targets["point_instance_label"], slots, jitter)
        out_targets["center_label"], out_targets["size_gts"] = boxes[..., :3].contiguous(), boxes[..., 3:].contiguous()
        out_targets["box_label_mask"] = mask
    return out_inputs, out_targets, scenes
