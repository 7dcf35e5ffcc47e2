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

What this package does not re-implement stays the reference's: the ``models`` stand-in keeps the
reference's ``models/`` directory on its ``__path__`` (``reference_root``, default: a ``models/ap_helper.py``
found on ``sys.path``), so ``from models import APCalculator, parse_predictions, parse_groundtruths``
(train_dist_mod.py:24) resolves lazily to the reference's own ``models/ap_helper.py`` instead of being shadowed.
"""
import importlib
import os
import sys
import types

# names train_dist_mod.py:24 takes from the package that live in the reference's models/ap_helper.py
_AP_HELPER_NAMES = ("APCalculator", "parse_predictions", "parse_groundtruths")


def _find_reference_models(reference_root):
    roots = [reference_root] if reference_root else list(sys.path) + [os.getcwd()]
    for r in roots:
        d = os.path.join(r or ".", "models")
        if os.path.isfile(os.path.join(d, "ap_helper.py")):
            return d
    return None


def install(scope="all", attention_backend="hip", reference_root=None, ops_binding="ctypes"):
    """``ops_binding``: which module becomes ``pointnet2._ext`` -- "ctypes" (butd_detr_amd.pointnet2_ext, the
    product path's binding) or "aten" (the compiled ATen / pybind11 front end of the same C ABI,
    butd_detr_amd/binding/pointnet2_aten.cpp; built on first use)."""
    if ops_binding == "aten":
        from .binding import build as _aten_build
        pointnet2_ext = _aten_build.load()
    elif ops_binding == "ctypes":
        from . import pointnet2_ext
    else:
        raise ValueError(ops_binding)
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
    ref_models = _find_reference_models(reference_root)
    models.__path__ = [ref_models] if ref_models else []

    def _fallthrough(name):   # PEP 562: only consulted for names the stand-in does not define
        if name in _AP_HELPER_NAMES or name == "ap_helper":
            if not models.__path__:
                raise ImportError(f"models.{name} is the reference's (models/ap_helper.py): pass "
                                  "reference_root= to dropin.install() or run from the reference root")
            mod = importlib.import_module("models.ap_helper")
            return mod if name == "ap_helper" else getattr(mod, name)
        raise AttributeError(f"module 'models' has no attribute {name!r}")

    models.__getattr__ = _fallthrough
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
