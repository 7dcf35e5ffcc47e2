"""Make the MI355X build importable under the reference's module names.

``install()`` registers this package's modules in ``sys.modules`` under the names the reference code
imports (train_dist_mod.py / models/*.py / pointnet2/*.py), so the reference driver runs unchanged on
top of the gfx950 kernels:

    import butd_detr_amd.dropin as dropin; dropin.install()
    from models import BeaUTyDETR            # -> butd_detr_amd.bdetr.BeaUTyDETR
    from models import HungarianMatcher, SetCriterion, compute_hungarian_loss   # -> butd_detr_amd.losses
    import pointnet2._ext as _ext            # -> butd_detr_amd.pointnet2_ext (the 9 pybind functions)
    import pointnet2_utils                   # -> butd_detr_amd.pointnet2_utils (Function API)

``scope='ops'`` registers only the operator layer (``pointnet2._ext``), leaving the reference's own
Python modules in charge -- the smallest possible swap.
"""
import sys
import types


def install(scope="all", attention_backend="hip"):
    from . import pointnet2_ext
    pkg = sys.modules.get("pointnet2")
    if pkg is None:
        pkg = types.ModuleType("pointnet2")
        pkg.__path__ = []
        sys.modules["pointnet2"] = pkg
    pkg._ext = pointnet2_ext
    sys.modules["pointnet2._ext"] = pointnet2_ext
    if scope == "ops":
        return
    from . import (attention_blocks, backbone_module, bdetr, encoder_decoder_layers, losses, modules,
                   pointnet2_modules, pointnet2_utils, pytorch_utils)
    for name, mod in (("pointnet2_utils", pointnet2_utils), ("pointnet2.pointnet2_utils", pointnet2_utils),
                      ("pointnet2_modules", pointnet2_modules), ("pointnet2.pointnet2_modules", pointnet2_modules),
                      ("pytorch_utils", pytorch_utils), ("pointnet2.pytorch_utils", pytorch_utils)):
        sys.modules[name] = mod
    pkg.pointnet2_utils = pointnet2_utils
    pkg.pointnet2_modules = pointnet2_modules
    models = types.ModuleType("models")
    models.__path__ = []
    models.BeaUTyDETR = bdetr.BeaUTyDETR
    # main_utils.py:27: `from models import HungarianMatcher, SetCriterion, compute_hungarian_loss`
    for name in ("HungarianMatcher", "SetCriterion", "compute_hungarian_loss", "generalized_box_iou3d",
                 "box_cxcyczwhd_to_xyzxyz"):
        setattr(models, name, getattr(losses, name))
    for name, mod in (("bdetr", bdetr), ("backbone_module", backbone_module),
                      ("encoder_decoder_layers", encoder_decoder_layers), ("modules", modules),
                      ("losses", losses)):
        setattr(models, name, mod)
        sys.modules[f"models.{name}"] = mod
    sys.modules["models"] = models
    if attention_backend:
        try:
            attention_blocks.set_backend(attention_backend)
        except Exception:
            if attention_backend == "hip":
                raise
