"""The operator API of `metrics` / `mm3d_pn2` on CPU tensors, backed by the CPU
oracle -- TEST INFRASTRUCTURE (never imported by the product).

Two users:
  * tests/golden/make_model_golden.py installs these as the `metrics` /
    `mm3d_pn2` modules the REFERENCE's completion/model_utils.py and
    models/*.py import (model_utils.py:19-21), so the reference's own Python
    runs in the build container and emits fixtures;
  * the CPU tests patch them over the op names of THIS repo's completion
    modules (`patch_ops`), so the repo's model code runs without a GPU and is
    compared with those fixtures.  The `-m gpu` tests compare the HIP path with
    the same fixtures.

Every callable keeps the reference wrapper's signature, dtypes and
differentiability (file:line cited per function).
"""
import numpy as np
import torch
from torch.autograd import Function

import oracle

OP_NAMES = ("furthest_point_sample", "furthest_point_sample_with_dist", "gather_points", "grouping_operation",
            "ball_query", "knn_op", "three_nn", "three_interpolate", "cd", "emd")


def _np(t):
    return np.ascontiguousarray(t.detach().cpu().numpy())


def _f(a, like):
    return torch.from_numpy(np.ascontiguousarray(a)).to(like.device)


class _Chamfer(Function):
    """chamfer_3DFunction (dist_chamfer_3D.py:26-64)."""

    @staticmethod
    def forward(ctx, xyz1, xyz2):
        d1, d2, i1, i2 = oracle.chamfer_forward(_np(xyz1), _np(xyz2))
        idx1, idx2 = _f(i1, xyz1), _f(i2, xyz1)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        return _f(d1, xyz1), _f(d2, xyz1), idx1, idx2

    @staticmethod
    def backward(ctx, g1, g2, _gi1, _gi2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        gx1, gx2 = oracle.chamfer_backward(_np(xyz1), _np(xyz2), _np(g1), _np(g2), _np(idx1), _np(idx2))
        return _f(gx1, xyz1), _f(gx2, xyz1)


class cd(torch.nn.Module):
    """chamfer_3DDist (dist_chamfer_3D.py:67-74)."""

    def forward(self, input1, input2):
        return _Chamfer.apply(input1.contiguous(), input2.contiguous())


class _Emd(Function):
    """emdFunction (emd_module.py:40-81): gradient to xyz1 only."""

    @staticmethod
    def forward(ctx, xyz1, xyz2, eps, iters):
        dist, assignment = oracle.emd_forward(_np(xyz1), _np(xyz2), eps, iters)
        ass = _f(assignment, xyz1)
        ctx.save_for_backward(xyz1, xyz2, ass)
        ctx.mark_non_differentiable(ass)
        return _f(dist, xyz1), ass

    @staticmethod
    def backward(ctx, gdist, _gass):
        xyz1, xyz2, ass = ctx.saved_tensors
        gx = oracle.emd_backward(_np(xyz1), _np(xyz2), _np(gdist), _np(ass))
        return _f(gx, xyz1), torch.zeros_like(xyz2), None, None


class emd(torch.nn.Module):
    """emdModule (emd_module.py:83-88)."""

    def forward(self, input1, input2, eps, iters):
        return _Emd.apply(input1.contiguous(), input2.contiguous(), eps, iters)


def furthest_point_sample(points_xyz, num_points):
    """FurthestPointSampling.apply (furthest_point_sample.py:7-36) -> (B, m) int32."""
    assert points_xyz.is_contiguous()
    return _f(oracle.furthest_point_sample(_np(points_xyz), int(num_points)), points_xyz)


def furthest_point_sample_with_dist(points_dist, num_points):
    """FurthestPointSamplingWithDist.apply (furthest_point_sample.py:42-70)."""
    assert points_dist.is_contiguous()
    return _f(oracle.furthest_point_sample_with_dist(_np(points_dist), int(num_points)), points_dist)


class _Gather(Function):
    """GatherPoints (gather_points.py:7-52)."""

    @staticmethod
    def forward(ctx, features, indices):
        assert features.is_contiguous() and indices.is_contiguous()
        ctx.save_for_backward(indices)
        ctx.n = features.size(2)
        return _f(oracle.gather_points(_np(features), _np(indices)), features)

    @staticmethod
    def backward(ctx, grad_out):
        idx, = ctx.saved_tensors
        return _f(oracle.gather_points_grad(_np(grad_out), _np(idx), ctx.n), grad_out), None


gather_points = _Gather.apply


class _Group(Function):
    """GroupingOperation (group_points.py:166-221)."""

    @staticmethod
    def forward(ctx, features, indices):
        assert features.is_contiguous() and indices.is_contiguous()
        ctx.save_for_backward(indices)
        ctx.n = features.size(2)
        return _f(oracle.grouping_operation(_np(features), _np(indices)), features)

    @staticmethod
    def backward(ctx, grad_out):
        idx, = ctx.saved_tensors
        return _f(oracle.grouping_operation_grad(_np(grad_out), _np(idx), ctx.n), grad_out), None


grouping_operation = _Group.apply


def ball_query(min_radius, max_radius, sample_num, xyz, center_xyz):
    """BallQuery.apply (ball_query.py:7-47) -> (B, M, S) int32."""
    assert xyz.is_contiguous() and center_xyz.is_contiguous()
    return _f(oracle.ball_query(float(min_radius), float(max_radius), int(sample_num), _np(xyz), _np(center_xyz)), xyz)


def knn_op(k, xyz, center_xyz=None, transposed=False):
    """KNN.apply (knn.py:7-72) -> (B, k, M) int32."""
    return _f(oracle.knn(int(k), _np(xyz), None if center_xyz is None else _np(center_xyz), transposed), xyz)


def three_nn(target, source):
    """ThreeNN.apply (three_nn.py:8-45) -> (sqrt(dist2) (B,N,3), idx (B,N,3) int32)."""
    assert target.is_contiguous() and source.is_contiguous()
    dist, idx = oracle.three_nn(_np(target), _np(source))
    return _f(dist, target), _f(idx, target)


class _Interp(Function):
    """ThreeInterpolate (three_interpolate.py:8-63): gradient to the features only."""

    @staticmethod
    def forward(ctx, features, indices, weight):
        assert features.is_contiguous() and indices.is_contiguous() and weight.is_contiguous()
        ctx.save_for_backward(indices, weight)
        ctx.m = features.size(2)
        return _f(oracle.three_interpolate(_np(features), _np(indices), _np(weight)), features)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        return _f(oracle.three_interpolate_grad(_np(grad_out), _np(idx), _np(weight), ctx.m), grad_out), None, None


three_interpolate = _Interp.apply


def patch_ops(monkeypatch, modules):
    """Replace, in every module of `modules`, each operator name it imported from `metrics` / `mm3d_pn2` with the
    oracle-backed callable of the same name (the product's own wrappers raise without a GPU)."""
    table = {name: globals()[name] for name in OP_NAMES}
    for mod in modules:
        for name, fn in table.items():
            if hasattr(mod, name):
                monkeypatch.setattr(mod, name, fn)
