"""TEST-INFRASTRUCTURE stand-in for the ``pointnet2._ext`` surface, backed by the CPU oracle.

Used (a) by tests/golden/make_golden.py to let the *unmodified reference Python* run on CPU (the
reference has no CPU path: pointnet2_utils.py:25-33 raises ImportError without ``_ext`` and every
C++ wrapper rejects CPU tensors), (b) by CPU-only tests of the host-side modules, which monkeypatch
it over ``butd_detr_amd.pointnet2_utils._ext``, and (c) by bench.py's ``cpu_baseline`` leg.
Never imported by the product package.
"""
import torch

from oracle import pointnet2_oracle as orc


def _np(t):
    return t.detach().cpu().numpy()


def furthest_point_sampling(points, nsamples):
    return torch.from_numpy(orc.furthest_point_sampling(_np(points), int(nsamples), multithread=True))


def gather_points(points, idx):
    return torch.from_numpy(orc.gather_points(_np(points), _np(idx)))


def gather_points_grad(grad_out, idx, n):
    return torch.from_numpy(orc.gather_points_grad(_np(grad_out), _np(idx), n))


def ball_query(new_xyz, xyz, radius, nsample):
    return torch.from_numpy(orc.ball_query(_np(new_xyz), _np(xyz), float(radius), int(nsample)))


def group_points(points, idx):
    return torch.from_numpy(orc.group_points(_np(points), _np(idx)))


def group_points_grad(grad_out, idx, n):
    return torch.from_numpy(orc.group_points_grad(_np(grad_out), _np(idx), n))


def three_nn(unknowns, knows):
    d, i = orc.three_nn(_np(unknowns), _np(knows))
    return [torch.from_numpy(d), torch.from_numpy(i)]


def three_interpolate(points, idx, weight):
    return torch.from_numpy(orc.three_interpolate(_np(points), _np(idx), _np(weight)))


def three_interpolate_grad(grad_out, idx, weight, m):
    return torch.from_numpy(orc.three_interpolate_grad(_np(grad_out), _np(idx), _np(weight), m))
