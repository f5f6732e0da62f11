"""Loader for oracle/_ref: the REFERENCE's own kernels compiled for gfx950 by oracle/build_ref_gpu.sh.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): loaded by tests/ (through tests/ref_kernels.py) and by bench.py's
baseline leg, never by the product.

Each function allocates what the reference's Python wrapper allocates (file:line cited) and calls the reference's
launcher / pybind function on CUDA tensors; results come back as torch tensors on the device.
`variant` "" = hipcc's default contraction (nvcc's -fmad=true counterpart), "_nofma" = -ffp-contract=off.
"""
import ctypes
import importlib.util
import os

import torch

REF_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")

_SYMS = {
    "fps": "_Z39furthest_point_sampling_kernel_launcheriiiPKfPfPiP12ihipStream_t",
    "fps_dist": "_Z49furthest_point_sampling_with_dist_kernel_launcheriiiPKfPfPiP12ihipStream_t",
    "ball_query": "_Z26ball_query_kernel_launcheriiiffiPKfS0_PiP12ihipStream_t",
    "knn": "_Z19knn_kernel_launcheriiiiPKfS0_PiPfP12ihipStream_t",
    "three_nn": "_Z24three_nn_kernel_launcheriiiPKfS0_PfPiP12ihipStream_t",
    "three_interpolate": "_Z33three_interpolate_kernel_launcheriiiiPKfPKiS0_PfP12ihipStream_t",
    "three_interpolate_grad": "_Z38three_interpolate_grad_kernel_launcheriiiiPKfPKiS0_PfP12ihipStream_t",
    "gather": "_Z29gather_points_kernel_launcheriiiiPKfPKiPfP12ihipStream_t",
    "gather_grad": "_Z34gather_points_grad_kernel_launcheriiiiPKfPKiPfP12ihipStream_t",
    "group": "_Z28group_points_kernel_launcheriiiiiPKfPKiPfP12ihipStream_t",
    "group_grad": "_Z33group_points_grad_kernel_launcheriiiiiPKfPKiPfP12ihipStream_t",
}

_cache = {}

SYNC = True   # tests compare results right away; a timing of many calls in a row sets this to False (the reference's own
              # wrappers do not synchronise either)


def _sync():
    if SYNC:
        torch.cuda.synchronize()


def available(variant=""):
    return all(os.path.exists(os.path.join(REF_DIR, f)) for f in
               (f"libref_pn2{variant}.so", f"ref_emd{variant}.so", f"ref_chamfer_3D{variant}.so"))


def _pn2(variant):
    key = ("pn2", variant)
    if key not in _cache:
        _cache[key] = ctypes.CDLL(os.path.join(REF_DIR, f"libref_pn2{variant}.so"))
    return _cache[key]


def _module(name, variant):
    key = (name, variant)
    if key not in _cache:
        full = f"{name}{variant}"
        spec = importlib.util.spec_from_file_location(full, os.path.join(REF_DIR, full + ".so"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _cache[key] = mod
    return _cache[key]


def _call(variant, name, *args):
    fn = getattr(_pn2(variant), _SYMS[name])
    fn.restype = None
    conv = []
    for a in args:
        if isinstance(a, torch.Tensor):
            assert a.is_cuda and a.is_contiguous()
            conv.append(ctypes.c_void_p(a.data_ptr()))
        elif isinstance(a, float):
            conv.append(ctypes.c_float(a))
        else:
            conv.append(ctypes.c_int(int(a)))
    conv.append(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    fn(*conv)
    _sync()


def fps(xyz, m, variant=""):
    """furthest_point_sample.py:29-33 (temp filled with 1e10, idx an uninitialised IntTensor)."""
    b, n, _ = xyz.shape
    temp = torch.full((b, n), 1e10, device=xyz.device, dtype=torch.float32)
    idx = torch.empty(b, m, device=xyz.device, dtype=torch.int32)
    _call(variant, "fps", b, n, m, xyz, temp, idx)
    return idx


def fps_with_dist(dist, m, variant=""):
    """furthest_point_sample.py:62-68."""
    b, n, _ = dist.shape
    temp = torch.full((b, n), 1e10, device=dist.device, dtype=torch.float32)
    idx = torch.empty(b, m, device=dist.device, dtype=torch.int32)
    _call(variant, "fps_dist", b, n, m, dist, temp, idx)
    return idx


def ball_query(min_radius, max_radius, nsample, xyz, center_xyz, variant=""):
    """ball_query.py:35-38 (idx zeroed)."""
    b, n, _ = xyz.shape
    m = center_xyz.shape[1]
    idx = torch.zeros(b, m, nsample, device=xyz.device, dtype=torch.int32)
    _call(variant, "ball_query", b, n, m, float(min_radius), float(max_radius), nsample, center_xyz, xyz, idx)
    return idx


def knn(k, xyz, center_xyz, variant=""):
    """knn.py:55-61: idx (B, npoint, k) zeroed, dist2 filled by the kernel; returns (idx, dist2) before the transpose."""
    b, n, _ = xyz.shape
    m = center_xyz.shape[1]
    idx = torch.zeros(b, m, k, device=xyz.device, dtype=torch.int32)
    dist2 = torch.zeros(b, m, k, device=xyz.device, dtype=torch.float32)
    _call(variant, "knn", b, n, m, k, xyz, center_xyz, idx, dist2)
    return idx, dist2


def three_nn(target, source, variant=""):
    """three_nn.py:31-35: returns (dist2, idx); the wrapper's sqrt is host-side glue, not repeated here."""
    b, n, _ = target.shape
    m = source.shape[1]
    dist2 = torch.empty(b, n, 3, device=target.device, dtype=torch.float32)
    idx = torch.empty(b, n, 3, device=target.device, dtype=torch.int32)
    _call(variant, "three_nn", b, n, m, target, source, dist2, idx)
    return dist2, idx


def three_interpolate(features, idx, weight, variant=""):
    """three_interpolate.py:32-35."""
    b, c, m = features.shape
    n = idx.shape[1]
    out = torch.empty(b, c, n, device=features.device, dtype=torch.float32)
    _call(variant, "three_interpolate", b, c, m, n, features, idx, weight, out)
    return out


def three_interpolate_grad(grad_out, idx, weight, m, variant=""):
    """three_interpolate.py:53-57 (grad_features zeroed)."""
    b, c, n = grad_out.shape
    g = torch.zeros(b, c, m, device=grad_out.device, dtype=torch.float32)
    _call(variant, "three_interpolate_grad", b, c, n, m, grad_out, idx, weight, g)
    return g


def gather_points(features, idx, variant=""):
    """gather_points.py:30-33."""
    b, c, n = features.shape
    npoint = idx.shape[1]
    out = torch.empty(b, c, npoint, device=features.device, dtype=torch.float32)
    _call(variant, "gather", b, c, n, npoint, features, idx, out)
    return out


def gather_points_grad(grad_out, idx, n, variant=""):
    """gather_points.py:44-48 (grad_features zeroed)."""
    b, c, npoint = grad_out.shape
    g = torch.zeros(b, c, n, device=grad_out.device, dtype=torch.float32)
    _call(variant, "gather_grad", b, c, n, npoint, grad_out, idx, g)
    return g


def grouping_operation(features, idx, variant=""):
    """group_points.py:190-194."""
    b, c, n = features.shape
    _, npoint, nsample = idx.shape
    out = torch.empty(b, c, npoint, nsample, device=features.device, dtype=torch.float32)
    _call(variant, "group", b, c, n, npoint, nsample, features, idx, out)
    return out


def grouping_operation_grad(grad_out, idx, n, variant=""):
    """group_points.py:213-217 (grad_features zeroed)."""
    b, c, npoint, nsample = grad_out.shape
    g = torch.zeros(b, c, n, device=grad_out.device, dtype=torch.float32)
    _call(variant, "group_grad", b, c, n, npoint, nsample, grad_out, idx, g)
    return g


def chamfer_forward(xyz1, xyz2, variant=""):
    """dist_chamfer_3D.py:33-44."""
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dev = xyz1.device
    d1 = torch.zeros(b, n, device=dev)
    d2 = torch.zeros(b, m, device=dev)
    i1 = torch.zeros(b, n, device=dev, dtype=torch.int32)
    i2 = torch.zeros(b, m, device=dev, dtype=torch.int32)
    _module("ref_chamfer_3D", variant).forward(xyz1, xyz2, d1, d2, i1, i2)
    _sync()
    return d1, d2, i1, i2


def chamfer_backward(xyz1, xyz2, g1, g2, i1, i2, variant=""):
    """dist_chamfer_3D.py:56-64."""
    gx1 = torch.zeros_like(xyz1)
    gx2 = torch.zeros_like(xyz2)
    _module("ref_chamfer_3D", variant).backward(xyz1, xyz2, gx1, gx2, g1, g2, i1, i2)
    _sync()
    return gx1, gx2


def emd_forward(xyz1, xyz2, eps, iters, variant=""):
    """emd_module.py:49-65: the work buffers exactly as emdFunction.forward allocates them."""
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dev = xyz1.device
    z = lambda *s, dt=torch.float32: torch.zeros(*s, device=dev, dtype=dt)
    dist = z(b, n)
    assignment = z(b, n, dt=torch.int32) - 1
    assignment_inv = z(b, m, dt=torch.int32) - 1
    price = z(b, m)
    bid = z(b, n, dt=torch.int32)
    bid_increments = z(b, n)
    max_increments = z(b, m)
    unass_idx = z(b * n, dt=torch.int32)
    max_idx = z(b * m, dt=torch.int32)
    unass_cnt = z(512, dt=torch.int32)
    unass_cnt_sum = z(512, dt=torch.int32)
    cnt_tmp = z(512, dt=torch.int32)
    _module("ref_emd", variant).forward(xyz1, xyz2, dist, assignment, price, assignment_inv, bid, bid_increments,
                                        max_increments, unass_idx, unass_cnt, unass_cnt_sum, cnt_tmp, max_idx, eps, iters)
    _sync()
    return dist, assignment, price


def emd_backward(xyz1, xyz2, graddist, assignment, variant=""):
    """emd_module.py:76-80 (only gradxyz1 is written)."""
    gx1 = torch.zeros_like(xyz1)
    gx2 = torch.zeros_like(xyz2)
    _module("ref_emd", variant).backward(xyz1, xyz2, gx1, graddist, assignment)
    _sync()
    return gx1, gx2
