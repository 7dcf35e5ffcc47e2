"""Generates tests/golden/resident_scans.pkl + resident_cases.npz by running the REFERENCE's dataset code
(/root/reference/src/joint_det_dataset.py: pickle_data, Joint3DDataset._augment / _get_target_boxes /
_get_detected_objects; src/visual_data_handlers.py: Scan.get_object_bbox) on three small synthetic scans.

    python tests/golden/make_resident_golden.py

The .pkl is written BY the reference's ``pickle_data`` from instances of the reference's own ``Scan`` class, i.e. it has
the on-disk format of ``{split}_v3scans.pkl`` (joint_det_dataset.py:96-99); the build reads it without the reference.
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden.make_augment_golden import REF, load_reference  # noqa: E402

N, SLOTS = 1500, 132
LABELS = ["chair", "table", "door", "couch", "cabinet", "shelf", "desk", "office chair", "bed", "pillow", "sink",
          "picture", "window", "toilet", "bookshelf", "monitor"]


def synthetic_scan(Scan, scan_id, seed):
    rng = np.random.RandomState(seed)
    scan = Scan.__new__(Scan)
    scan.scan_id, scan.top_scan_dir, scan.choices, scan.semantic_label_idx = scan_id, "", None, None
    scan.pc = rng.uniform(-4, 4, (N, 3))
    scan.pc[:, 2] = rng.uniform(0, 3, N)
    scan.orig_pc = np.copy(scan.pc)
    scan.color = rng.uniform(0, 1, (N, 3)).astype(np.float32)
    objs = []
    for k in range(10 + seed % 5):
        centre = scan.pc[rng.randint(N)]
        near = np.argsort(np.linalg.norm(scan.pc - centre, axis=1))[: rng.randint(20, 160)]
        objs.append({"object_id": k, "points": np.array(sorted(near)), "instance_label": LABELS[rng.randint(len(LABELS))]})
    # two objects that share points (a point of two targets keeps the LAST target's slot, :507-508)
    objs.append({"object_id": len(objs), "points": objs[0]["points"][::2].copy(), "instance_label": "chair"})
    scan.three_d_objects = objs
    return scan


class FakeSelf:
    mean_rgb = np.array([109.8, 97.2, 83.8]) / 256
    split, augment, detect_intermediate, augment_det = "train", True, False, False


def main():
    mod = load_reference()
    from src.visual_data_handlers import Scan
    os.chdir(REF)
    self = FakeSelf()
    self.label_map = mod.read_label_mapping("data/meta_data/scannetv2-labels.combined.tsv", label_from="raw_category",
                                            label_to="id")
    scans = {f"scene{7000 + i:04d}_00": synthetic_scan(Scan, f"scene{7000 + i:04d}_00", 11 + i) for i in range(3)}
    mod.pickle_data(os.path.join(HERE, "resident_scans.pkl"), scans)
    tmp = tempfile.mkdtemp()
    self.data_path = tmp + "/"
    os.makedirs(f"{tmp}/group_free_pred_bboxes_train")
    out = {}
    for i, (sid, scan) in enumerate(scans.items()):
        rng = np.random.RandomState(50 + i)
        k = 9 + 3 * i
        lo = rng.uniform(-3, 2, (k, 3))
        det = {"box": np.concatenate([lo, lo + rng.uniform(0.2, 1.5, (k, 3))], 1), "class": [LABELS[j % len(LABELS)] for j in range(k)],
               "logits": rng.randn(k, 485).astype(np.float32)}
        np.save(f"{tmp}/group_free_pred_bboxes_train/{sid}.npy", det, allow_pickle=True)
        out[f"det_box_{i}"] = det["box"]
        out[f"det_class_ids_{i}"] = np.array([mod.DC.nyu40id2class[self.label_map[c]] for c in det["class"]])
        out[f"obj_class_ids_{i}"] = np.array([mod.DC.nyu40id2class[self.label_map[o["instance_label"]]]
                                              for o in scan.three_d_objects])
    cases = [(0, 3, True, 21), (1, [2, 5, 0, len(scans["scene7001_00"].three_d_objects) - 1], True, 22),
             (2, [0, len(scans["scene7002_00"].three_d_objects) - 1, 4], False, 23), (0, [1], False, 24)]
    from oracle import augment_oracle
    for c, (si, tids, rotate, seed) in enumerate(cases):
        sid = list(scans)[si]
        scan = scans[sid]
        scan.pc = np.copy(scan.orig_pc)                                           # joint_det_dataset.py:633
        np.random.seed(seed)
        color = scan.color - self.mean_rgb                                        # :416
        pc, color, aug = mod.Joint3DDataset._augment(self, scan.pc, color, rotate)
        scan.pc = pc                                                              # :443
        state = np.random.get_state()
        n_t = 1 if isinstance(tids, int) else len(tids)
        jitter = 0.95 + 0.1 * np.random.random((n_t, 6))                          # what :516 is about to draw
        np.random.set_state(state)
        boxes, mask, label = mod.Joint3DDataset._get_target_boxes(self, {"target_id": tids}, scan)
        det_boxes, det_mask, det_cls, _ = mod.Joint3DDataset._get_detected_objects(self, "train", sid, aug)
        state = np.random.get_state()
        all_jitter = 0.95 + 0.1 * np.random.random((SLOTS, 6))                    # what :553-554 is about to draw
        np.random.set_state(state)
        cls_ids, all_boxes, all_mask = mod.Joint3DDataset._get_scene_objects(self, scan)
        np.random.seed(seed)
        replay = augment_oracle.draw(rotate, N, True, np.random)
        assert np.array_equal(replay["noise"], aug["noise"]) and replay["scale"] == aug["scale"]
        out.update({f"c{c}_scan": np.asarray(si), f"c{c}_tids": np.asarray([tids] if isinstance(tids, int) else tids),
                    f"c{c}_single": np.asarray(int(isinstance(tids, int))), f"c{c}_rotate": np.asarray(int(rotate)),
                    f"c{c}_theta": np.asarray([aug["theta_z"], aug["theta_x"], aug["theta_y"]]),
                    f"c{c}_flips": np.asarray([int(aug.get("yz_flip", False)), int(aug.get("xz_flip", False))]),
                    f"c{c}_shift": aug["shift"].reshape(3), f"c{c}_scale": np.asarray(aug["scale"]),
                    f"c{c}_noise": aug["noise"], f"c{c}_color_gain": replay["color_gain"], f"c{c}_jitter": jitter,
                    f"c{c}_out_pc": pc, f"c{c}_out_color": color, f"c{c}_boxes": boxes, f"c{c}_mask": mask,
                    f"c{c}_label": label, f"c{c}_det_boxes": det_boxes, f"c{c}_det_mask": det_mask, f"c{c}_det_cls": det_cls,
                    f"c{c}_all_jitter": all_jitter, f"c{c}_all_boxes": all_boxes, f"c{c}_all_mask": all_mask,
                    f"c{c}_all_cls": cls_ids})
        print(c, sid, tids, "targets labelled:", [(label == t).sum() for t in range(n_t)])
    out["n_cases"] = np.asarray(len(cases))
    np.savez_compressed(os.path.join(HERE, "resident_cases.npz"), **out)


if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("the reference is not mounted here")
    main()
