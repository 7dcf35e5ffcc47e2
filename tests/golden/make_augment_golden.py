"""Generates tests/golden/augment_*.npz by running the REFERENCE's augmentation code
(/root/reference/src/joint_det_dataset.py: Joint3DDataset._augment, rot_x/y/z, box2points, points2box; the
detected-box and target-box snippets are driven through the same module functions) on seeded inputs.  Heavy
imports of that module that the image lacks (h5py, wandb, ...) are stubbed: none is touched by these functions.

    python tests/golden/make_augment_golden.py
"""
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def load_reference():
    for name in ("h5py", "wandb", "ipdb", "termcolor", "plyfile"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                stub = types.ModuleType(name)
                stub.PlyData = stub.PlyElement = object      # names imported at module level, never used here
                stub.set_trace = lambda *a, **k: None
                stub.colored = lambda s, *a, **k: s
                sys.modules[name] = stub
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        return importlib.import_module("src.joint_det_dataset")
    finally:
        os.chdir(cwd)


class FakeSelf:
    mean_rgb = np.array([109.8, 97.2, 83.8]) / 256


def case(mod, name, seed, rotate, n=700, d=132, n_det=17, n_targets=5):
    rng = np.random.RandomState(1000 + seed)
    pc = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
    color = (rng.uniform(0, 1, (n, 3)) - FakeSelf.mean_rgb).astype(np.float32)
    det = np.zeros((d, 6))
    det[:n_det, :3] = rng.uniform(-3, 3, (n_det, 3))
    det[:n_det, 3:] = rng.uniform(0.2, 2, (n_det, 3))
    instance = -np.ones(n, dtype=np.int64)
    for t in range(n_targets):
        instance[rng.choice(n, 30, replace=False)] = t
    np.random.seed(seed)
    out_pc, out_color, aug = mod.Joint3DDataset._augment(FakeSelf(), pc.copy(), color.copy(), rotate)
    color_gain_state = None
    # the colour gains are not returned by _augment: replay the stream to capture them
    np.random.seed(seed)
    from oracle import augment_oracle
    replay = augment_oracle.draw(rotate, n, True, np.random)
    assert np.array_equal(replay["noise"], aug["noise"]) and replay["scale"] == aug["scale"]
    # detected boxes, joint_det_dataset.py:595-607 through the reference's helpers
    pts = mod.box2points(det).reshape(-1, 3)
    pts = mod.rot_z(pts, aug["theta_z"])
    pts = mod.rot_x(pts, aug["theta_x"])
    pts = mod.rot_y(pts, aug["theta_y"])
    if aug.get("yz_flip", False):
        pts[:, 0] = -pts[:, 0]
    if aug.get("xz_flip", False):
        pts[:, 1] = -pts[:, 1]
    pts += aug["shift"]
    pts *= aug["scale"]
    out_det = mod.points2box(pts.reshape(-1, 8, 3))
    # target boxes, :497-522 with visual_data_handlers.py:245-258 (Scan._set_axis_align_bbox)
    from src.visual_data_handlers import Scan
    jitter = 0.95 + 0.1 * np.random.random((n_targets, 6))
    boxes = np.zeros((d, 6))
    boxes[:n_targets] = np.stack([Scan._set_axis_align_bbox(out_pc[instance == t]).reshape(-1)
                                  for t in range(n_targets)])
    boxes = np.concatenate(((boxes[:, :3] + boxes[:, 3:]) * 0.5, boxes[:, 3:] - boxes[:, :3]), 1)
    boxes[:n_targets] *= jitter
    boxes[n_targets:, :3] = 1000
    np.savez_compressed(
        os.path.join(HERE, f"augment_{name}.npz"), in_pc=pc, in_color=color, in_det=det, in_instance=instance,
        rotate=np.asarray(int(rotate)), seed=np.asarray(seed), theta=np.asarray([aug["theta_z"], aug["theta_x"], aug["theta_y"]]),
        flips=np.asarray([int(aug.get("yz_flip", False)), int(aug.get("xz_flip", False))]), shift=aug["shift"].reshape(3),
        scale=np.asarray(aug["scale"]), noise=aug["noise"], color_gain=replay["color_gain"], jitter=jitter,
        out_pc=out_pc, out_color=out_color, out_det=out_det, out_boxes=boxes, n_targets=np.asarray(n_targets))
    print(name, "theta", aug["theta_z"], aug["theta_x"], aug["theta_y"], "flips", aug.get("yz_flip"), aug.get("xz_flip"))


def main():
    if not os.path.exists(REF):
        sys.exit("the reference is not mounted here")
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    mod = load_reference()
    case(mod, "rotate_a", 3, True)
    case(mod, "rotate_b", 4, True)
    case(mod, "view_dependent", 5, False)


if __name__ == "__main__":
    main()
