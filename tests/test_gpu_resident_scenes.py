"""-m gpu: scenes resident in HBM + box bookkeeping on the device (butd_detr_amd/resident_scenes.py, include/butd_augment.h:
butd_object_boxes) against samples produced by the reference's dataset code -- ``_augment``, ``_get_target_boxes`` (incl. a
point that belongs to two target objects), ``_get_detected_objects`` -- on scans stored in the reference's own pickle
format (tests/golden/make_resident_golden.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def store():
    from butd_detr_amd.resident_scenes import ResidentScenes, read_scans
    z = np.load(os.path.join(HERE, "golden", "resident_cases.npz"))
    scans = read_scans(os.path.join(HERE, "golden", "resident_scans.pkl"))
    detected = {sid: {"box": z[f"det_box_{i}"], "class_ids": z[f"det_class_ids_{i}"]} for i, sid in enumerate(scans)}
    classes = {sid: z[f"obj_class_ids_{i}"] for i, sid in enumerate(scans)}
    return ResidentScenes.from_scans(scans, "cuda", detected=detected, object_class_ids=classes), list(scans), z


def test_batches_reproduce_the_reference_samples(store):
    rs, ids, z = store
    assert rs.cloud.shape == (3, 1500, 6) and rs.cloud.is_cuda and rs.bytes() > 0
    for c in range(int(z["n_cases"])):
        g = lambda k: z[f"c{c}_{k}"]
        sid = ids[int(g("scan"))]
        tids = g("tids").tolist()
        target = tids[0] if int(g("single")) else tids
        scene = {"theta_z": float(g("theta")[0]), "theta_x": float(g("theta")[1]), "theta_y": float(g("theta")[2]),
                 "yz_flip": bool(g("flips")[0]), "xz_flip": bool(g("flips")[1]), "shift": g("shift"), "scale": float(g("scale"))}
        jitter = np.ones((1, 132, 6))
        jitter[0, :len(tids)] = g("jitter")
        inputs, targets, _ = rs.batch([sid], [target], augment=True, rotate=bool(g("rotate")), scene_params=[scene],
                                      jitter=jitter, noise=torch.from_numpy(g("noise")[None]),
                                      color_gain=torch.from_numpy(g("color_gain")[None]))
        pc = inputs["point_clouds"].cpu().numpy()[0]
        # the reference transforms float64 coordinates (``align_to_axes`` promotes them, visual_data_handlers.py:185-191)
        # and rounds the sample to float32 at the end (:744); the resident cloud is their float32 rounding: 2.4e-7 m
        # of input rounding through a rotation and a scale -> within 2 micrometres
        np.testing.assert_allclose(pc[:, :3], g("out_pc"), rtol=0, atol=2e-6)
        np.testing.assert_allclose(pc[:, 3:], g("out_color"), rtol=0, atol=1e-7)
        np.testing.assert_array_equal(targets["point_instance_label"].cpu().numpy()[0], g("label").astype(np.int64))
        boxes = torch.cat([targets["center_label"], targets["size_gts"]], -1).cpu().numpy()[0]
        np.testing.assert_allclose(boxes, g("boxes"), rtol=2e-6, atol=2e-6)      # (hull of float32 points vs float64)
        np.testing.assert_array_equal(targets["box_label_mask"].cpu().numpy()[0], g("mask").astype(np.float32))
        np.testing.assert_allclose(inputs["det_boxes"].cpu().numpy()[0], g("det_boxes"), rtol=1e-5, atol=1e-5)
        np.testing.assert_array_equal(inputs["det_bbox_label_mask"].cpu().numpy()[0], g("det_mask"))
        np.testing.assert_array_equal(inputs["det_class_ids"].cpu().numpy()[0], g("det_cls").astype(np.int64))
        # _get_scene_objects (:524-560): every object's box from the augmented cloud, jittered; zeros in unused slots
        all_boxes, all_mask, all_cls = rs.scene_objects([sid], point_clouds=inputs["point_clouds"],
                                                        jitter=g("all_jitter")[None])
        np.testing.assert_allclose(all_boxes.cpu().numpy()[0], g("all_boxes"), rtol=2e-6, atol=2e-6)
        np.testing.assert_array_equal(all_mask.cpu().numpy()[0], g("all_mask"))
        np.testing.assert_array_equal(all_cls.cpu().numpy()[0], g("all_cls").astype(np.int64))


def test_batch_of_several_scenes_feeds_the_model_shapes_and_eval_mode_is_the_stored_scene(store):
    rs, ids, z = store
    inputs, targets, scenes = rs.batch([ids[2], ids[0], ids[2]], [[1, 2], 0, 3], augment=True,
                                       rotate=[True, False, True], rng=np.random.RandomState(3), seed=5)
    assert inputs["point_clouds"].shape == (3, 1500, 6) and inputs["det_boxes"].shape == (3, 132, 6)
    assert inputs["det_bbox_label_mask"].dtype == torch.bool and inputs["det_class_ids"].dtype == torch.int64
    assert targets["box_label_mask"].sum(1).tolist() == [2.0, 1.0, 1.0] and len(scenes) == 3
    assert not torch.equal(inputs["point_clouds"][0], inputs["point_clouds"][2])          # own transform per sample
    again = rs.batch([ids[2], ids[0], ids[2]], [[1, 2], 0, 3], augment=True, rotate=[True, False, True],
                     rng=np.random.RandomState(3), seed=5)
    assert torch.equal(again[0]["point_clouds"], inputs["point_clouds"])                   # deterministic in (rng, seed)
    ev_in, ev_t, _ = rs.batch([ids[1]], [4], augment=False)
    assert torch.equal(ev_in["point_clouds"][0], rs.cloud[1])
    pts = rs.cloud[1][torch.from_numpy(np.asarray(read_points(rs, 1, 4))).long().cuda(), :3]
    want_c = (pts.max(0).values + pts.min(0).values) / 2
    assert torch.allclose(ev_t["center_label"][0, 0], want_c, atol=1e-6)
    assert float(ev_t["center_label"][0, 1, 0]) == 1000.0                                  # padding slot (:518)
    with pytest.raises(IndexError):
        rs.batch([ids[0]], [99])


def read_points(rs, scene, obj):
    ptr = rs.obj_ptr[scene].cpu().numpy()
    return rs.obj_points[ptr[obj]:ptr[obj + 1]].cpu().numpy()
