"""The reference's on-disk scene format (``{split}_v3scans.pkl``: ``pickle_data`` of a ``{scan_id: Scan}`` dict,
joint_det_dataset.py:96-99,1019-1034) read without the reference's classes.  The fixture was written by the
reference's own ``pickle_data`` from instances of its ``Scan`` class (tests/golden/make_resident_golden.py)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PKL = os.path.join(HERE, "golden", "resident_scans.pkl")


def test_reads_the_reference_pickle_without_the_reference():
    assert not any(m == "src" or m.startswith("src.") for m in sys.modules), "the reference must not be importable here"
    from butd_detr_amd.resident_scenes import corners_to_center_size, read_scans
    scans = read_scans(PKL)
    assert sorted(scans) == ["scene7000_00", "scene7001_00", "scene7002_00"]
    for sid, scan in scans.items():
        assert scan.scan_id == sid and scan.orig_pc.shape == (1500, 3) and scan.color.shape == (1500, 3)
        assert np.array_equal(scan.pc, scan.orig_pc) and 0.0 <= scan.color.min() and scan.color.max() < 1.0
        assert len(scan.three_d_objects) >= 11
        for k, o in enumerate(scan.three_d_objects):
            assert o["object_id"] == k and isinstance(o["instance_label"], str)
            assert o["points"].min() >= 0 and o["points"].max() < 1500
    z = np.load(os.path.join(HERE, "golden", "resident_cases.npz"))
    cs = corners_to_center_size(z["det_box_0"])
    np.testing.assert_allclose(cs[:, :3] - cs[:, 3:] / 2, z["det_box_0"][:, :3], atol=1e-12)
    np.testing.assert_allclose(cs[:, :3] + cs[:, 3:] / 2, z["det_box_0"][:, 3:], atol=1e-12)
