"""Builds ``pointnet2_aten.cpp`` -- the ATen / pybind11 front end of the C ABI under the reference's module
surface (``pointnet2._ext``, bindings.cpp:11-24) -- in-tree with torch.utils.cpp_extension:

    python -m butd_detr_amd.binding.build          ->  butd_detr_amd/binding/_build/butd_pointnet2_aten.so

Plain C++ (no device code): it links libbutd_detr_hip.so (``python -m butd_detr_amd.build``) and torch.
The product path does not need it (it binds the same C ABI with ctypes); it exists so that the boundary is
proven to bind where the reference's own shims (src/sampling.cpp:9-18 ...) would call it.
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
BUILD_DIR = os.path.join(HERE, "_build")
NAME = "butd_pointnet2_aten"


def load(verbose=False):
    """Compile (first call) and import the extension module."""
    import torch  # noqa: F401
    from torch.utils import cpp_extension
    from .. import build as libbuild
    lib_path = libbuild.build()
    lib_dir = os.path.dirname(lib_path)
    os.makedirs(BUILD_DIR, exist_ok=True)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    return cpp_extension.load(
        name=NAME, sources=[os.path.join(HERE, "pointnet2_aten.cpp")],
        extra_include_paths=[os.path.join(ROOT, "include"), os.path.join(rocm, "include")],
        extra_cflags=["-O2", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1"],
        extra_ldflags=[f"-L{lib_dir}", "-lbutd_detr_hip", f"-Wl,-rpath,{lib_dir}",
                       f"-L{os.path.join(os.path.dirname(torch.__file__), 'lib')}", "-lc10_hip", "-ltorch_hip"],
        build_directory=BUILD_DIR, with_cuda=False, verbose=verbose)


if __name__ == "__main__":
    m = load(verbose=True)
    print(m.__file__)
