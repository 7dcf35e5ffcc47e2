"""ctypes/numpy front-end of oracle/pointnet2_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, bench.py's ``cpu_baseline`` leg and ``__graft_entry__.smoke()`` may import this
module.  The product package (``butd_detr_amd``) never does: it fails loudly when its HIP
library is missing instead of falling back to this code.

Argument/return conventions mirror the reference C++ wrappers
(pointnet2/_ext_src/src/{sampling,ball_query,group_points,interpolate}.cpp): outputs are
allocated here with the same initial values (zeros; FPS scratch = 1e10).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpointnet2_oracle.so")
_lib = None

_F = ctypes.POINTER(ctypes.c_float)
_I = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    """Compile the C restatement with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, "pointnet2_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libpointnet2_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


def host_cores():
    """Usable cores: affinity mask capped by the cgroup v2 CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_opt_n_threads.restype = ctypes.c_int
        _lib.oracle_opt_n_threads.argtypes = [ctypes.c_int]
        _lib.oracle_set_num_threads(ctypes.c_int(host_cores()))
    return _lib


def set_num_threads(n):
    lib().oracle_set_num_threads(ctypes.c_int(int(n)))


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_F)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_I)


def opt_n_threads(work_size):
    return int(lib().oracle_opt_n_threads(int(work_size)))


def furthest_point_sampling(points, nsamples, multithread=False):
    """points (B,N,3) f32 -> (B,nsamples) i32.  sampling.cpp:70-91."""
    points, pp = _f(points)
    b, n, _ = points.shape
    out = np.zeros((b, nsamples), dtype=np.int32)
    tmp = np.full((b, n), 1e10, dtype=np.float32)
    fn = lib().oracle_furthest_point_sampling_mt if multithread else lib().oracle_furthest_point_sampling
    fn(ctypes.c_int(b), ctypes.c_int(n), ctypes.c_int(nsamples), pp,
       tmp.ctypes.data_as(_F), out.ctypes.data_as(_I))
    return out


def gather_points(points, idx):
    """points (B,C,N), idx (B,M) -> (B,C,M).  sampling.cpp:20-44."""
    points, pp = _f(points)
    idx, ip = _i(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.zeros((b, c, m), dtype=np.float32)
    lib().oracle_gather_points(b, c, n, m, pp, ip, out.ctypes.data_as(_F))
    return out


def gather_points_grad(grad_out, idx, n):
    """grad_out (B,C,M), idx (B,M) -> (B,C,n).  sampling.cpp:46-69."""
    grad_out, gp = _f(grad_out)
    idx, ip = _i(idx)
    b, c, m = grad_out.shape
    out = np.zeros((b, c, n), dtype=np.float32)
    lib().oracle_gather_points_grad(b, c, int(n), m, gp, ip, out.ctypes.data_as(_F))
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    """new_xyz (B,M,3), xyz (B,N,3) -> (B,M,nsample) i32.  ball_query.cpp:13-37."""
    new_xyz, qp = _f(new_xyz)
    xyz, pp = _f(xyz)
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    out = np.zeros((b, m, nsample), dtype=np.int32)
    lib().oracle_ball_query(b, n, m, ctypes.c_float(radius), int(nsample), qp, pp,
                            out.ctypes.data_as(_I))
    return out


def group_points(points, idx):
    """points (B,C,N), idx (B,M,S) -> (B,C,M,S).  group_points.cpp:17-40."""
    points, pp = _f(points)
    idx, ip = _i(idx)
    b, c, n = points.shape
    _, m, s = idx.shape
    out = np.zeros((b, c, m, s), dtype=np.float32)
    lib().oracle_group_points(b, c, n, m, s, pp, ip, out.ctypes.data_as(_F))
    return out


def group_points_grad(grad_out, idx, n):
    """grad_out (B,C,M,S), idx (B,M,S) -> (B,C,n).  group_points.cpp:42-65."""
    grad_out, gp = _f(grad_out)
    idx, ip = _i(idx)
    b, c, m, s = grad_out.shape
    out = np.zeros((b, c, n), dtype=np.float32)
    lib().oracle_group_points_grad(b, c, int(n), m, s, gp, ip, out.ctypes.data_as(_F))
    return out


def three_nn(unknown, known):
    """unknown (B,n,3), known (B,m,3) -> dist2 (B,n,3) f32, idx (B,n,3) i32.  interpolate.cpp:19-48."""
    unknown, up = _f(unknown)
    known, kp = _f(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    dist2 = np.zeros((b, n, 3), dtype=np.float32)
    idx = np.zeros((b, n, 3), dtype=np.int32)
    lib().oracle_three_nn(b, n, m, up, kp, dist2.ctypes.data_as(_F), idx.ctypes.data_as(_I))
    return dist2, idx


def three_interpolate(points, idx, weight):
    """points (B,c,m), idx/weight (B,n,3) -> (B,c,n).  interpolate.cpp:50-77."""
    points, pp = _f(points)
    idx, ip = _i(idx)
    weight, wp = _f(weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = np.zeros((b, c, n), dtype=np.float32)
    lib().oracle_three_interpolate(b, c, m, n, pp, ip, wp, out.ctypes.data_as(_F))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    """grad_out (B,c,n), idx/weight (B,n,3) -> (B,c,m).  interpolate.cpp:79-104."""
    grad_out, gp = _f(grad_out)
    idx, ip = _i(idx)
    weight, wp = _f(weight)
    b, c, n = grad_out.shape
    out = np.zeros((b, c, int(m)), dtype=np.float32)
    lib().oracle_three_interpolate_grad(b, c, n, int(m), gp, ip, wp, out.ctypes.data_as(_F))
    return out
