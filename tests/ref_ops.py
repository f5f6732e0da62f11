"""The operator API of `metrics` / `mm3d_pn2` on GPU tensors, backed by the REFERENCE'S OWN KERNELS (oracle/_ref through
oracle/ref_gpu.py) -- TEST INFRASTRUCTURE, the GPU twin of tests/oracle_ops.py: same names, signatures and differentiability as
the reference's wrappers (file:line in oracle_ops.py).  Used by tests/report_reference_model_step.py to time a network step
of this repo's model code on the reference's kernels."""
import torch
from torch.autograd import Function

from oracle import ref_gpu as ref

OP_NAMES = ("furthest_point_sample", "furthest_point_sample_with_dist", "gather_points", "grouping_operation",
            "ball_query", "knn_op", "three_nn", "three_interpolate", "cd", "emd")


class _Chamfer(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        d1, d2, i1, i2 = ref.chamfer_forward(xyz1, xyz2)
        ctx.save_for_backward(xyz1, xyz2, i1, i2)
        ctx.mark_non_differentiable(i1, i2)
        return d1, d2, i1, i2

    @staticmethod
    def backward(ctx, g1, g2, _a, _b):
        xyz1, xyz2, i1, i2 = ctx.saved_tensors
        return ref.chamfer_backward(xyz1, xyz2, g1.contiguous(), g2.contiguous(), i1, i2)


class cd(torch.nn.Module):
    def forward(self, a, b):
        return _Chamfer.apply(a.contiguous().float(), b.contiguous().float())


class _Emd(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2, eps, iters):
        dist, ass, _ = ref.emd_forward(xyz1, xyz2, eps, iters)
        ctx.save_for_backward(xyz1, xyz2, ass)
        ctx.mark_non_differentiable(ass)
        return dist, ass

    @staticmethod
    def backward(ctx, gdist, _g):
        xyz1, xyz2, ass = ctx.saved_tensors
        gx1, gx2 = ref.emd_backward(xyz1, xyz2, gdist.contiguous(), ass)
        return gx1, gx2, None, None


class emd(torch.nn.Module):
    def forward(self, a, b, eps, iters):
        return _Emd.apply(a.contiguous().float(), b.contiguous().float(), eps, iters)


def furthest_point_sample(points_xyz, num_points):
    return ref.fps(points_xyz.contiguous(), int(num_points))


def furthest_point_sample_with_dist(points_dist, num_points):
    return ref.fps_with_dist(points_dist.contiguous(), int(num_points))


class _Gather(Function):
    @staticmethod
    def forward(ctx, features, indices):
        ctx.save_for_backward(indices)
        ctx.n = features.size(2)
        return ref.gather_points(features.contiguous(), indices.contiguous())

    @staticmethod
    def backward(ctx, grad_out):
        idx, = ctx.saved_tensors
        return ref.gather_points_grad(grad_out.contiguous(), idx, ctx.n), None


gather_points = _Gather.apply


class _Group(Function):
    @staticmethod
    def forward(ctx, features, indices):
        ctx.save_for_backward(indices)
        ctx.n = features.size(2)
        return ref.grouping_operation(features.contiguous(), indices.contiguous())

    @staticmethod
    def backward(ctx, grad_out):
        idx, = ctx.saved_tensors
        return ref.grouping_operation_grad(grad_out.contiguous(), idx, ctx.n), None


grouping_operation = _Group.apply


def ball_query(min_radius, max_radius, sample_num, xyz, center_xyz):
    return ref.ball_query(float(min_radius), float(max_radius), int(sample_num), xyz.contiguous(), center_xyz.contiguous())


def knn_op(k, xyz, center_xyz=None, transposed=False):
    """KNN.apply (knn.py:17-66) -> (B, k, M) int32."""
    if center_xyz is None:
        center_xyz = xyz
    if transposed:
        xyz, center_xyz = xyz.transpose(2, 1).contiguous(), center_xyz.transpose(2, 1).contiguous()
    idx, _ = ref.knn(int(k), xyz.contiguous(), center_xyz.contiguous())
    return idx.transpose(2, 1).contiguous()


def three_nn(target, source):
    dist2, idx = ref.three_nn(target.contiguous(), source.contiguous())
    return torch.sqrt(dist2), idx


class _Interp(Function):
    @staticmethod
    def forward(ctx, features, indices, weight):
        ctx.save_for_backward(indices, weight)
        ctx.m = features.size(2)
        return ref.three_interpolate(features.contiguous(), indices.contiguous(), weight.contiguous())

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        return ref.three_interpolate_grad(grad_out.contiguous(), idx, weight, ctx.m), None, None


three_interpolate = _Interp.apply


def patch_ops(modules):
    """Replace the operator names of `modules` with the reference-kernel callables; returns a function that undoes it."""
    saved = []
    for mod in modules:
        for name in OP_NAMES:
            if hasattr(mod, name):
                saved.append((mod, name, getattr(mod, name)))
                setattr(mod, name, globals()[name])

    def undo():
        for mod, name, fn in saved:
            setattr(mod, name, fn)
    return undo
