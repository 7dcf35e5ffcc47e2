"""CPU (-m "not gpu"): the augmentation oracle (oracle/augment_oracle.py) against vectors captured from the
reference's own code (tests/golden/make_augment_golden.py), incl. that its `draw` replays numpy's random stream
exactly as `Joint3DDataset._augment` consumes it."""
import glob
import os

import numpy as np
import pytest

from oracle import augment_oracle as ao

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "augment_*.npz")))


def params_of(z):
    return {"theta_z": float(z["theta"][0]), "theta_x": float(z["theta"][1]), "theta_y": float(z["theta"][2]),
            "yz_flip": bool(z["flips"][0]), "xz_flip": bool(z["flips"][1]), "shift": z["shift"][None, :],
            "scale": float(z["scale"]), "noise": z["noise"], "color_gain": z["color_gain"]}


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[8:-4] for p in GOLD])
def test_oracle_reproduces_the_reference_augmentation(path):
    z = np.load(path)
    a = params_of(z)
    pc, color = ao.augment_points(z["in_pc"], z["in_color"], a)
    np.testing.assert_array_equal(pc, z["out_pc"])            # same dtypes, same roundings: bit for bit
    np.testing.assert_array_equal(color, z["out_color"])
    np.testing.assert_allclose(ao.augment_boxes(z["in_det"], a), z["out_det"], rtol=0, atol=1e-12)
    slots = z["in_det"].shape[0]
    jitter = np.ones((slots, 6))
    jitter[: int(z["n_targets"])] = z["jitter"]
    boxes, mask = ao.instance_boxes(z["out_pc"], z["in_instance"], slots, jitter)
    np.testing.assert_allclose(boxes, z["out_boxes"], rtol=0, atol=1e-12)
    assert mask.sum() == int(z["n_targets"])


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[8:-4] for p in GOLD])
def test_draw_replays_the_reference_random_stream(path):
    z = np.load(path)
    np.random.seed(int(z["seed"]))
    a = ao.draw(bool(z["rotate"]), z["in_pc"].shape[0], True, np.random)
    want = params_of(z)
    for k in ("theta_z", "theta_x", "theta_y", "scale", "yz_flip", "xz_flip"):
        assert a[k] == want[k], k
    np.testing.assert_array_equal(a["noise"], want["noise"])
    np.testing.assert_array_equal(a["shift"], want["shift"])
    np.testing.assert_array_equal(a["color_gain"], want["color_gain"])


def test_device_augment_has_no_cpu_fallback():
    import torch
    from butd_detr_amd import device_augment as da
    params = da.pack_params([da.draw_scene(True, np.random.RandomState(0))], "cpu")
    assert params.numel() == 256
    with pytest.raises(RuntimeError, match="CPU not supported"):
        da.augment_points(torch.zeros(1, 4, 6), params)
