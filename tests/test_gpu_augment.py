"""-m gpu: device-side augmentation (include/butd_augment.h, butd_detr_amd/device_augment.py) against vectors
captured from the reference's dataset code (tests/golden/make_augment_golden.py) and the numpy oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import augment_oracle as ao  # noqa: E402
from tests.test_augment_cpu import GOLD, params_of  # noqa: E402


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[8:-4] for p in GOLD])
def test_kernels_reproduce_the_reference_sample(path):
    from butd_detr_amd import device_augment as da
    z = np.load(path)
    a = params_of(z)
    scene = dict(a, shift=a["shift"].reshape(3))
    params = da.pack_params([scene], "cuda")
    cloud = torch.from_numpy(np.concatenate([z["in_pc"], z["in_color"]], 1)[None]).cuda()
    out = da.augment_points(cloud, params, noise=torch.from_numpy(z["noise"][None]),
                            color_gain=torch.from_numpy(z["color_gain"][None])).cpu().numpy()[0]
    np.testing.assert_array_max_ulp(out[:, :3], z["out_pc"], maxulp=1)      # BLAS may contract where we do not
    assert (out[:, :3] == z["out_pc"]).mean() > 0.999
    np.testing.assert_array_equal(out[:, 3:], z["out_color"])
    det = da.augment_boxes(torch.from_numpy(z["in_det"][None].astype(np.float32)).cuda(), params).cpu().numpy()[0]
    np.testing.assert_allclose(det, ao.augment_boxes(z["in_det"].astype(np.float32), a).astype(np.float32),
                               rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(det, z["out_det"], rtol=1e-5, atol=1e-5)     # golden from float64 input boxes
    slots = z["in_det"].shape[0]
    jitter = np.ones((1, slots, 6))
    jitter[0, : int(z["n_targets"])] = z["jitter"]
    aug_cloud = torch.from_numpy(np.concatenate([z["out_pc"], z["out_color"]], 1)[None]).cuda()
    boxes, mask = da.instance_boxes(aug_cloud, torch.from_numpy(z["in_instance"][None]).cuda(), slots,
                                    torch.from_numpy(jitter))
    np.testing.assert_array_equal(boxes.cpu().numpy()[0], z["out_boxes"].astype(np.float32))
    assert mask.cpu().numpy()[0].tolist() == [1.0] * int(z["n_targets"]) + [0.0] * (slots - int(z["n_targets"]))


def test_batch_api_and_device_side_draws():
    """augment_batch on the bench-shaped batch: every scene gets its own transform, targets are recomputed from
    the augmented points, the hash-drawn noise / colour gains have the reference's ranges, and the call is
    deterministic in (host rng, seed)."""
    from butd_detr_amd import device_augment as da
    from butd_detr_amd.train_step import synthetic_batch
    inputs, targets = synthetic_batch(4, torch.device("cuda"), n_points=20000, tokens=16)
    outs = [da.augment_batch(inputs, targets, rotate=[True, False, True, True], rng=np.random.RandomState(5), seed=9)
            for _ in range(2)]
    (i0, t0, scenes), (i1, t1, _) = outs
    assert torch.equal(i0["point_clouds"], i1["point_clouds"]) and torch.equal(t0["center_label"], t1["center_label"])
    pc_in, pc_out = inputs["point_clouds"].cpu().numpy(), i0["point_clouds"].cpu().numpy()
    for b, a in enumerate(scenes):
        full = dict(a, shift=np.asarray(a["shift"])[None, :], noise=np.zeros((pc_in.shape[1], 3)),
                    color_gain=np.ones((pc_in.shape[1], 3)))
        want, _ = ao.augment_points(pc_in[b, :, :3], None, full)
        resid = pc_out[b, :, :3] / np.float32(a["scale"]) - want / np.float32(a["scale"])     # = noise
        assert resid.min() > -1e-5 and resid.max() < 5e-3 + 1e-5 and 2e-3 < resid.mean() < 3e-3
        gain = (pc_out[b, :, 3:6] + np.float32(ao.MEAN_RGB)) / (pc_in[b, :, 3:6] + np.float32(ao.MEAN_RGB))
        assert gain.min() > 0.98 - 1e-3 and gain.max() < 1.02 + 1e-3
    # target boxes follow the augmented cloud
    valid = t0["box_label_mask"][0] > 0
    assert valid.sum() >= 1
    lab = targets["point_instance_label"][0].cpu().numpy()
    t = 0
    sel = pc_out[0, lab == t, :3]
    c = t0["center_label"][0, t].cpu().numpy()
    assert np.all(c > sel.min(0) * 0.9 - 0.2) and np.all(c < sel.max(0) * 1.1 + 0.2)
    assert float(t0["center_label"][0, -1, 0]) == 1000.0       # padding slots (joint_det_dataset.py:518)
    assert i0["det_boxes"].shape == inputs["det_boxes"].shape
