"""CPU parity oracle for the MVP point-cloud op layer -- TEST INFRASTRUCTURE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; the product path (``mvp_benchmark_amd``) never
does and fails loudly when its HIP library is missing.

``libmvp_oracle.so`` is the C restatement in ``mvp_oracle.c`` (each function
cites the reference ``.cu`` lines it follows); this module is a thin NumPy
front-end that allocates outputs/scratch exactly as the reference's Python
wrappers do (zero / -1 / 1e10 initial values, file:line cited per function).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmvp_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)
_i64p = ctypes.POINTER(ctypes.c_longlong)


def build(force=False):
    """Compile libmvp_oracle.so with the Makefile next to this file."""
    src = os.path.join(_HERE, "mvp_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libmvp_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_num_threads.restype = ctypes.c_int
    return _lib


def set_contraction(on):
    """True (default): the canonical fma chain; False: every product and sum rounded on its own (mvp_oracle.c)."""
    lib().orc_set_contraction(1 if on else 0)


def num_threads():
    return int(lib().orc_num_threads())


def set_num_threads(t):
    lib().orc_set_num_threads(int(t))


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _pf(a):
    return a.ctypes.data_as(_f32p)


def _pi(a):
    return a.ctypes.data_as(_i32p)


def _check(rc, name):
    if rc != 0:
        raise RuntimeError("oracle %s failed with code %d" % (name, rc))


# ---------------------------------------------------------------- chamfer
def chamfer_forward(xyz1, xyz2):
    """chamfer_3DFunction.forward, dist_chamfer_3D.py:28-47 (zero-filled
    outputs) -> (dist1, dist2, idx1, idx2)."""
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist1 = np.zeros((b, n), np.float32)
    dist2 = np.zeros((b, m), np.float32)
    idx1 = np.zeros((b, n), np.int32)
    idx2 = np.zeros((b, m), np.int32)
    _check(lib().orc_chamfer_forward(b, n, m, _pf(xyz1), _pf(xyz2), _pf(dist1),
                                     _pf(dist2), _pi(idx1), _pi(idx2)),
           "chamfer_forward")
    return dist1, dist2, idx1, idx2


def chamfer_backward(xyz1, xyz2, graddist1, graddist2, idx1, idx2):
    """chamfer_3DFunction.backward, dist_chamfer_3D.py:50-64."""
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    g1, g2, idx1, idx2 = _f(graddist1), _f(graddist2), _i(idx1), _i(idx2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    gx1 = np.zeros_like(xyz1)
    gx2 = np.zeros_like(xyz2)
    _check(lib().orc_chamfer_backward(b, n, m, _pf(xyz1), _pf(xyz2), _pf(gx1),
                                      _pf(gx2), _pf(g1), _pf(g2), _pi(idx1),
                                      _pi(idx2)), "chamfer_backward")
    return gx1, gx2


# -------------------------------------------------------------------- emd
def emd_forward(xyz1, xyz2, eps, iters, return_stats=False):
    """emdFunction.forward, emd_module.py:42-70 -> (dist, assignment)."""
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, _ = xyz1.shape
    assert xyz2.shape[1] == n and xyz2.shape[0] == b
    dist = np.zeros((b, n), np.float32)
    assignment = np.zeros((b, n), np.int32) - 1
    stats = np.zeros((b, 2), np.int64)
    fn = lib().orc_emd_forward
    fn.argtypes = [ctypes.c_int, ctypes.c_int, _f32p, _f32p, _f32p, _i32p,
                   ctypes.c_float, ctypes.c_int, _i64p]
    _check(fn(b, n, _pf(xyz1), _pf(xyz2), _pf(dist), _pi(assignment),
              ctypes.c_float(eps), int(iters), stats.ctypes.data_as(_i64p)),
           "emd_forward")
    if return_stats:
        return dist, assignment, stats
    return dist, assignment


def emd_forward_ex(xyz1, xyz2, eps, iters, getmax_lowest=False):
    """emd_forward with the GetMax schedule choice exposed and the number of
    unassigned persons at the start of every round -> (dist, assignment,
    stats (b,2), trace (b,iters))."""
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, _ = xyz1.shape
    assert xyz2.shape[1] == n and xyz2.shape[0] == b
    dist = np.zeros((b, n), np.float32)
    assignment = np.zeros((b, n), np.int32) - 1
    stats = np.zeros((b, 2), np.int64)
    trace = np.zeros((b, int(iters)), np.int32)
    fn = lib().orc_emd_forward_ex
    fn.argtypes = [ctypes.c_int, ctypes.c_int, _f32p, _f32p, _f32p, _i32p,
                   ctypes.c_float, ctypes.c_int, _i64p, ctypes.c_int, _i32p]
    _check(fn(b, n, _pf(xyz1), _pf(xyz2), _pf(dist), _pi(assignment),
              ctypes.c_float(eps), int(iters), stats.ctypes.data_as(_i64p),
              int(bool(getmax_lowest)), _pi(trace)), "emd_forward_ex")
    return dist, assignment, stats, trace


def emd_backward(xyz1, xyz2, graddist, assignment):
    """emdFunction.backward, emd_module.py:73-81 (gradient to xyz1 only)."""
    xyz1, xyz2, g, a = _f(xyz1), _f(xyz2), _f(graddist), _i(assignment)
    b, n, _ = xyz1.shape
    gx = np.zeros_like(xyz1)
    _check(lib().orc_emd_backward(b, n, _pf(xyz1), _pf(xyz2), _pf(gx), _pf(g),
                                  _pi(a)), "emd_backward")
    return gx


# -------------------------------------------------------------------- fps
def fps_block_size(n):
    return int(lib().orc_fps_block_size(int(n)))


def furthest_point_sample(xyz, m):
    """FurthestPointSampling.forward, furthest_point_sample.py:17-36."""
    xyz = _f(xyz)
    b, n, _ = xyz.shape
    temp = np.full((b, n), 1e10, np.float32)
    idx = np.zeros((b, m), np.int32)
    _check(lib().orc_furthest_point_sampling(b, n, int(m), _pf(xyz), _pf(temp),
                                             _pi(idx)), "fps")
    return idx


def furthest_point_sample_with_dist(dist, m):
    """FurthestPointSamplingWithDist.forward, furthest_point_sample.py:49-70."""
    dist = _f(dist)
    b, n, _ = dist.shape
    temp = np.full((b, n), 1e10, np.float32)
    idx = np.zeros((b, m), np.int32)
    _check(lib().orc_furthest_point_sampling_with_dist(
        b, n, int(m), _pf(dist), _pf(temp), _pi(idx)), "fps_with_dist")
    return idx


# ------------------------------------------------------------------ query
def ball_query(min_radius, max_radius, sample_num, xyz, center_xyz):
    """BallQuery.forward, ball_query.py:15-43 (idx zero-initialised :35)."""
    xyz, center = _f(xyz), _f(center_xyz)
    b, n, _ = xyz.shape
    m = center.shape[1]
    idx = np.zeros((b, m, sample_num), np.int32)
    fn = lib().orc_ball_query
    fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                   ctypes.c_float, ctypes.c_int, _f32p, _f32p, _i32p]
    _check(fn(b, n, m, min_radius, max_radius, int(sample_num), _pf(center),
              _pf(xyz), _pi(idx)), "ball_query")
    return idx


def knn(k, xyz, center_xyz=None, transposed=False, return_dist=False):
    """KNN.forward, knn.py:17-66 -> idx (B, k, npoint)."""
    xyz = _f(xyz)
    center = xyz if center_xyz is None else _f(center_xyz)
    if transposed:
        xyz = np.ascontiguousarray(xyz.transpose(0, 2, 1))
        center = np.ascontiguousarray(center.transpose(0, 2, 1))
    b, n, _ = xyz.shape
    m = center.shape[1]
    idx = np.zeros((b, m, k), np.int32)
    dist2 = np.zeros((b, m, k), np.float32)
    _check(lib().orc_knn(b, n, m, int(k), _pf(xyz), _pf(center), _pi(idx),
                         _pf(dist2)), "knn")
    idx_t = np.ascontiguousarray(idx.transpose(0, 2, 1))
    if return_dist:
        return idx_t, dist2
    return idx_t


def three_nn(target, source):
    """ThreeNN.forward, three_nn.py:17-39 -> (sqrt(dist2), idx)."""
    target, source = _f(target), _f(source)
    b, n, _ = target.shape
    m = source.shape[1]
    dist2 = np.zeros((b, n, 3), np.float32)
    idx = np.zeros((b, n, 3), np.int32)
    _check(lib().orc_three_nn(b, n, m, _pf(target), _pf(source), _pf(dist2),
                              _pi(idx)), "three_nn")
    return np.sqrt(dist2), idx


# ----------------------------------------------------- interpolate/gather
def three_interpolate(features, indices, weight):
    """ThreeInterpolate.forward, three_interpolate.py:17-40."""
    f, idx, w = _f(features), _i(indices), _f(weight)
    b, c, m = f.shape
    n = idx.shape[1]
    out = np.zeros((b, c, n), np.float32)
    _check(lib().orc_three_interpolate(b, c, m, n, _pf(f), _pi(idx), _pf(w),
                                       _pf(out)), "three_interpolate")
    return out


def three_interpolate_grad(grad_out, indices, weight, m):
    """ThreeInterpolate.backward, three_interpolate.py:43-63."""
    g, idx, w = _f(grad_out), _i(indices), _f(weight)
    b, c, n = g.shape
    gp = np.zeros((b, c, m), np.float32)
    _check(lib().orc_three_interpolate_grad(b, c, n, int(m), _pf(g), _pi(idx),
                                            _pf(w), _pf(gp)),
           "three_interpolate_grad")
    return gp


def gather_points(features, indices):
    """GatherPoints.forward, gather_points.py:14-36."""
    f, idx = _f(features), _i(indices)
    b, c, n = f.shape
    npoint = idx.shape[1]
    out = np.zeros((b, c, npoint), np.float32)
    _check(lib().orc_gather_points(b, c, n, npoint, _pf(f), _pi(idx), _pf(out)),
           "gather_points")
    return out


def gather_points_grad(grad_out, indices, n):
    """GatherPoints.backward, gather_points.py:39-50."""
    g, idx = _f(grad_out), _i(indices)
    b, c, npoint = g.shape
    gp = np.zeros((b, c, n), np.float32)
    _check(lib().orc_gather_points_grad(b, c, int(n), npoint, _pf(g), _pi(idx),
                                        _pf(gp)), "gather_points_grad")
    return gp


def grouping_operation(features, indices):
    """GroupingOperation.forward, group_points.py:172-196."""
    f, idx = _f(features), _i(indices)
    b, c, n = f.shape
    _, npoint, nsample = idx.shape
    out = np.zeros((b, c, npoint, nsample), np.float32)
    _check(lib().orc_group_points(b, c, n, npoint, nsample, _pf(f), _pi(idx),
                                  _pf(out)), "group_points")
    return out


def grouping_operation_grad(grad_out, indices, n):
    """GroupingOperation.backward, group_points.py:199-218."""
    g, idx = _f(grad_out), _i(indices)
    b, c, npoint, nsample = g.shape
    gp = np.zeros((b, c, n), np.float32)
    _check(lib().orc_group_points_grad(b, c, int(n), npoint, nsample, _pf(g),
                                       _pi(idx), _pf(gp)), "group_points_grad")
    return gp
