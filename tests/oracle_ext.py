"""TEST-ONLY alias of ``oracle.ext_adapter`` (the ``pointnet2._ext`` surface backed by the CPU oracle),
kept for the CPU tests that monkeypatch it over ``butd_detr_amd.pointnet2_utils._ext``."""
from oracle.ext_adapter import *  # noqa: F401,F403
from oracle import ext_adapter as _impl

globals().update({k: getattr(_impl, k) for k in dir(_impl) if not k.startswith("__")})
