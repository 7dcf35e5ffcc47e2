"""Scenes that stay in HBM across epochs, with the per-sample box bookkeeping on the device
(reference: src/joint_det_dataset.py, src/visual_data_handlers.py).

The reference loads ``{split}_v3scans.pkl`` -- a pickled ``{scan_id: Scan}`` dict (joint_det_dataset.py:96-99, written by
``pickle_data``, :1028-1034) -- into host memory, and every ``__getitem__`` (:626-790) copies the scene's 50 000 points,
augments them in numpy on a DataLoader worker, walks ``scan.three_d_objects[tid]['points']`` to label the target points
and to box them (:497-522), transforms the 132 detected boxes (:562-607), and the batch then crosses PCIe.  Here

* ``read_scans`` reads that file format WITHOUT the reference's classes (its instances are plain attribute bags);
* ``ResidentScenes`` uploads every scene ONCE: the clouds ``[xyz | colour - mean]`` as one (S, N, 6) tensor (1 201
  ScanNet scenes x 50 000 points = 1.4 GB of the 288 GB), all objects' point lists as one CSR array, the detected boxes;
* ``batch(...)`` builds the model's ``inputs`` and the criterion's box targets for a list of (scan id, target object ids)
  with a row gather, the augmentation kernels of include/butd_augment.h and ``butd_object_boxes`` -- nothing per point
  happens on the host; it draws the same few scalars per scene as the reference, in its order.

What stays host-side string work (and is passed in as ready arrays by the caller): utterances, token spans /
``positive_map``, class-name -> id tables (data/model_util_scannet.py, scannetv2-labels.combined.tsv).
"""
import pickle

import numpy as np
import torch

from . import _hiplib, device_augment

MAX_NUM_OBJ = 132            # joint_det_dataset.py:33


class _Record:
    """Attribute bag standing in for the reference's ``Scan`` / ``ScanNetMappings`` instances while unpickling."""


class _ReferenceUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split(".")[0] in ("src", "visual_data_handlers", "joint_det_dataset"):
            return _Record
        return super().find_class(module, name)


def read_scans(path):
    """``{scan_id: scan}`` from a file written by the reference's ``pickle_data(filename, all_scans)``
    (joint_det_dataset.py:1019-1034: an item count, then the items): ``scan.pc`` / ``scan.orig_pc`` (N, 3),
    ``scan.color`` (N, 3) in [0, 1), ``scan.three_d_objects`` = [{'object_id', 'points', 'instance_label'}, ...]."""
    with open(path, "rb") as f:
        up = _ReferenceUnpickler(f)
        count = up.load()
        items = [up.load() for _ in range(count)]
    if not items or not isinstance(items[0], dict):
        raise ValueError(f"{path}: not a pickle_data() file of a scan dict")
    return items[0]


def corners_to_center_size(boxes):
    """(K, 6) xmin ymin zmin xmax ymax zmax -> centre + size (joint_det_dataset.py:584-587)."""
    boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 6)
    return np.concatenate(((boxes[:, :3] + boxes[:, 3:]) * 0.5, boxes[:, 3:] - boxes[:, :3]), 1)


class ResidentScenes:
    def __init__(self, device, slots=MAX_NUM_OBJ, mean_rgb=device_augment.MEAN_RGB):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("ResidentScenes: CPU not supported (the scenes live in HBM)")
        self.slots, self.mean_rgb = slots, np.asarray(mean_rgb, dtype=np.float64)
        self._pending, self.index = [], {}
        self.cloud = None

    # ------------------------------------------------------------------ filling
    def add_scan(self, scan_id, pc, color, objects, detected=None, object_class_ids=None):
        """One scene: ``pc`` (N, 3) axis-aligned coordinates (``scan.orig_pc``), ``color`` (N, 3) in [0, 1),
        ``objects`` = ``scan.three_d_objects``; ``detected``: {'box': (K, 6) corners, 'class_ids': (K,) int,
        optional 'logits'} -- the detector's boxes of ``group_free_pred_bboxes_{split}/{scan_id}.npy`` with its class
        NAMES already mapped to ids (joint_det_dataset.py:573-582).  ``object_class_ids``: class id of every object
        (``DC.nyu40id2class[label_map[instance_label]]`` or 325, :536-541), only needed by ``scene_objects``."""
        if self.cloud is not None:
            raise RuntimeError("ResidentScenes.add_scan after finalize()")
        pc = np.asarray(pc, dtype=np.float64)
        cloud = np.concatenate([pc, np.asarray(color, dtype=np.float64) - self.mean_rgb], 1).astype(np.float32)
        points = [np.asarray(o["points"], dtype=np.int32).reshape(-1) for o in objects]
        det = None
        if detected is not None:
            k = len(detected["class_ids"])
            if k >= self.slots:
                raise ValueError(f"{scan_id}: {k} detected boxes do not fit {self.slots} slots")   # :589
            box = np.zeros((self.slots, 6), np.float32)
            box[:k] = corners_to_center_size(detected["box"])
            mask = np.zeros(self.slots, bool)
            mask[:k] = True
            cls = np.zeros(self.slots, np.int64)
            cls[:k] = np.asarray(detected["class_ids"], dtype=np.int64)
            det = (box, mask, cls)
        cls_ids = np.zeros(self.slots, np.int64)
        if object_class_ids is not None:
            k = min(len(points), self.slots)
            cls_ids[:k] = np.asarray(object_class_ids, dtype=np.int64)[:k]
        self.index[scan_id] = len(self._pending)
        self._pending.append((cloud, points, det, cls_ids))

    @classmethod
    def from_scans(cls, scans, device, detected=None, object_class_ids=None, **kw):
        """``scans``: {scan_id: Scan-like} (``read_scans``); ``detected``: {scan_id: dict} as for ``add_scan``."""
        store = cls(device, **kw)
        for scan_id, scan in scans.items():
            pc = getattr(scan, "orig_pc", None)
            store.add_scan(scan_id, scan.pc if pc is None else pc, scan.color, scan.three_d_objects,
                           None if detected is None else detected.get(scan_id),
                           None if object_class_ids is None else object_class_ids.get(scan_id))
        return store.finalize()

    def finalize(self):
        n = {c.shape[0] for c, _, _, _ in self._pending}
        if len(n) != 1:
            raise ValueError(f"scenes of different sizes {sorted(n)}: the reference keeps 50 000 points per scene "
                             "(visual_data_handlers.py:111-118)")
        self.n_points = n.pop()
        dev = self.device
        self.cloud = torch.from_numpy(np.stack([c for c, _, _, _ in self._pending])).to(dev)          # (S, N, 6)
        self.max_objects = max(len(p) for _, p, _, _ in self._pending)
        ptr = np.zeros((len(self._pending), self.max_objects + 1), np.int64)
        flat, off = [], 0
        for s, (_, points, _, _) in enumerate(self._pending):
            for k, p in enumerate(points):
                ptr[s, k] = off
                flat.append(p)
                off += p.size
            ptr[s, len(points):] = off
        self.n_objects = np.asarray([len(p) for _, p, _, _ in self._pending])
        self.obj_ptr = torch.from_numpy(ptr).to(dev)
        self.obj_points = torch.from_numpy(np.concatenate(flat) if flat else np.zeros(0, np.int32)).to(dev)
        self.has_detected = all(d is not None for _, _, d, _ in self._pending)
        if self.has_detected:
            self.det_boxes = torch.from_numpy(np.stack([d[0] for _, _, d, _ in self._pending])).to(dev)
            self.det_mask = torch.from_numpy(np.stack([d[1] for _, _, d, _ in self._pending])).to(dev)
            self.det_class = torch.from_numpy(np.stack([d[2] for _, _, d, _ in self._pending])).to(dev)
        self.obj_class = torch.from_numpy(np.stack([c for _, _, _, c in self._pending])).to(dev)
        self._pending = None
        return self

    def scene_objects(self, scan_ids, point_clouds=None, jitter=None):
        """``_get_scene_objects`` (joint_det_dataset.py:524-560): the boxes of ALL objects of every scene (first 132),
        from ``point_clouds`` (B, N, >= 3) -- the batch's augmented clouds -- or the stored ones -> ``all_bboxes``
        (B, 132, 6) centre + size (zeros in unused slots), ``all_bbox_label_mask`` (B, 132) bool, ``all_class_ids``
        (B, 132).  ``jitter`` (B, 132, 6): the training split's ``0.95 + 0.1 * random`` (:553-554)."""
        dev, B, G = self.device, len(scan_ids), self.slots
        rows = [self.index[s] for s in scan_ids]
        scene = torch.tensor(rows, dtype=torch.int32, device=dev)
        pc = self.cloud.index_select(0, scene.long()) if point_clouds is None else point_clouds.contiguous().float()
        tids = np.full((B, G), -1, np.int32)
        for b, r in enumerate(rows):
            k = min(int(self.n_objects[r]), G)
            tids[b, :k] = np.arange(k)
        t_dev = torch.from_numpy(tids).to(dev)
        jit = None if jitter is None else torch.as_tensor(np.asarray(jitter, dtype=np.float64)).to(dev)
        scratch = torch.empty((B, G, 6), dtype=torch.int32, device=dev)
        boxes, mask = torch.empty((B, G, 6), device=dev), torch.empty((B, G), device=dev)
        lib = _hiplib.load()
        with torch.cuda.device(dev):
            err = lib.butd_object_boxes(B, pc.shape[1], pc.shape[-1], G, scene.data_ptr(), self.obj_ptr.data_ptr(),
                                        self.obj_ptr.shape[1], self.obj_points.data_ptr(), t_dev.data_ptr(),
                                        pc.data_ptr(), None if jit is None else jit.data_ptr(), None,
                                        scratch.data_ptr(), boxes.data_ptr(), mask.data_ptr(),
                                        torch.cuda.current_stream(dev).cuda_stream)
        _hiplib.check(err, "butd_object_boxes")
        keep = t_dev >= 0                                    # :531-533: every listed object is kept
        return boxes * keep[..., None], keep, self.obj_class.index_select(0, scene.long()) * keep

    def bytes(self):
        return sum(t.numel() * t.element_size() for t in (self.cloud, self.obj_ptr, self.obj_points))

    # ------------------------------------------------------------------ one batch
    def batch(self, scan_ids, target_ids, *, augment=True, rotate=True, rng=np.random, seed=0, noise=None,
              color_gain=None, scene_params=None, jitter=None):
        """``scan_ids``: B scene ids; ``target_ids``: per sample an object id or a list of them (``anno['target_id']``,
        joint_det_dataset.py:500-505).  -> (inputs, targets) on the device: ``point_clouds`` (B, N, 6), ``det_boxes``
        (B, 132, 6), ``det_bbox_label_mask``, ``det_class_ids``; ``center_label``, ``size_gts``, ``box_label_mask``,
        ``point_instance_label``.  ``augment``: the training-split transform (:358-403) with per-scene draws from ``rng``
        in the reference's order (``device_augment.draw_scene``; then the target-box jitter, :516); ``rotate``: bool or
        per-sample list (:432-441).  ``noise`` / ``color_gain`` (B, N, 3): impose the per-point draws (tests);
        otherwise they come from the device hash seeded with ``seed``.  ``scene_params`` (list of ``draw_scene``-style
        dicts) / ``jitter`` (B, slots, 6) impose the host draws as well."""
        if self.cloud is None:
            raise RuntimeError("ResidentScenes.batch before finalize()")
        dev, B, G, N = self.device, len(scan_ids), self.slots, self.n_points
        rows = [self.index[s] for s in scan_ids]
        tids = np.full((B, G), -1, np.int32)
        for b, t in enumerate(target_ids):
            t = [t] if isinstance(t, (int, np.integer)) else list(t)[:G]
            if any(k < 0 or k >= self.n_objects[rows[b]] for k in t):
                raise IndexError(f"{scan_ids[b]}: target object id out of range")
            tids[b, :len(t)] = t
        scene = torch.tensor(rows, dtype=torch.int32, device=dev)
        pc = self.cloud.index_select(0, scene.long())                                   # (B, N, 6) row gather
        inputs, scenes = {}, None
        if not augment:
            jitter = None
        if augment:
            flags = rotate if hasattr(rotate, "__len__") else [rotate] * B
            scenes = scene_params or [device_augment.draw_scene(bool(f), rng) for f in flags]
            params = device_augment.pack_params(scenes, dev)
            pc = device_augment.augment_points(pc, params, noise=noise, color_gain=color_gain, seed=seed)
            if jitter is None:
                jitter = np.ones((B, G, 6))
                for b in range(B):                                                      # :516, per sample
                    k = int((tids[b] >= 0).sum())
                    jitter[b, :k] = 0.95 + 0.1 * rng.random((k, 6))
            jitter = torch.as_tensor(np.asarray(jitter, dtype=np.float64)).to(dev)
        inputs["point_clouds"] = pc
        if self.has_detected:
            boxes = self.det_boxes.index_select(0, scene.long())
            inputs["det_boxes"] = device_augment.augment_boxes(boxes, params) if augment else boxes
            inputs["det_bbox_label_mask"] = self.det_mask.index_select(0, scene.long())
            inputs["det_class_ids"] = self.det_class.index_select(0, scene.long())
        label = torch.full((B, N), -1, dtype=torch.int64, device=dev)
        scratch = torch.empty((B, G, 6), dtype=torch.int32, device=dev)
        center_size = torch.empty((B, G, 6), device=dev)
        mask = torch.empty((B, G), device=dev)
        t_dev = torch.from_numpy(tids).to(dev)
        lib = _hiplib.load()
        with torch.cuda.device(dev):
            err = lib.butd_object_boxes(B, N, pc.shape[-1], G, scene.data_ptr(), self.obj_ptr.data_ptr(),
                                        self.obj_ptr.shape[1], self.obj_points.data_ptr(), t_dev.data_ptr(),
                                        pc.data_ptr(), None if jitter is None else jitter.data_ptr(),
                                        label.data_ptr(), scratch.data_ptr(), center_size.data_ptr(), mask.data_ptr(),
                                        torch.cuda.current_stream(dev).cuda_stream)
        _hiplib.check(err, "butd_object_boxes")
        targets = {"center_label": center_size[..., :3].contiguous(), "size_gts": center_size[..., 3:].contiguous(),
                   "box_label_mask": mask, "point_instance_label": label}
        return inputs, targets, scenes
