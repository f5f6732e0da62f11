"""autograd.Function front-ends of the PointNet++ set-abstraction operators.

One module for the whole op family (the reference spreads them over
`ops/<op>/<op>.py`, one pybind extension each; those import paths are kept as
thin re-exports).  Each Function keeps the reference's class name, positional
`apply` signature, output dtype/shape and autograd contract, allocates its
outputs on the input's device and hands raw device pointers to libmvpops
through `_lib.call` (C ABI: include/mvpops.h).

Reference wrappers mirrored here (paths under utils/mm3d_pn2/ops/):
  furthest_point_sample/furthest_point_sample.py:7-78
  ball_query/ball_query.py:7-47        knn/knn.py:7-72
  interpolate/three_nn.py:8-45         interpolate/three_interpolate.py:8-63
  gather_points/gather_points.py:7-52  group_points/group_points.py:166-221
"""
import threading

import torch
from torch.autograd import Function

from .._lib import call, fps_scratch_bytes, knn_scratch_bytes, scatter_scratch_bytes


def _need_contiguous(*tensors):
    for t in tensors:
        assert t.is_contiguous()


def _new(ref, *shape, dtype=torch.float32, zero=False):
    make = torch.zeros if zero else torch.empty
    return make(*shape, dtype=dtype, device=ref.device)


SCATTER_OVERWRITE, SCATTER_INDEX_READY = 1, 2      # mode bits of the *_grad_ws entry points (mvpops.h)

# Inverted index lists of the scatter-add gradients, kept per index tensor: the same
# neighbour graph is differentiated through several gathers per step (keys and values of
# SA_module, centre + neighbour features, xyz and features of edge_preserve_sampling), and a
# retained graph may be differentiated repeatedly.  An entry holds a reference to its index
# (and weight) tensor, so the storage it describes cannot be recycled while the entry lives;
# `_version` detects in-place edits made through PyTorch.
#
# Contract: an index / weight tensor that went through a backward pass must not be rewritten
# through a RAW POINTER afterwards (`_lib.call` writing into a pre-allocated index workspace,
# `.data` edits): `_version` does not see such writes and the next backward on it would reduce
# over the stale inverted list.  Code that recycles index buffers that way calls
# `clear_scatter_cache()` after the rewrite, or switches the cache off
# (`SCATTER_CACHE = False`: every backward rebuilds its list, ~15 us).
# An entry is published only AFTER the kernel that builds its list has been launched
# successfully (`_scatter_commit`), so a call that raises leaves nothing behind.  Autograd runs
# one thread per device: the list is guarded by a lock.
SCATTER_CACHE = True
_TRANSPOSED = []
_TRANSPOSED_CAP = 12
_TRANSPOSED_LOCK = threading.Lock()


def clear_scatter_cache():
    """Forget every cached inverted index list (see the contract above)."""
    with _TRANSPOSED_LOCK:
        del _TRANSPOSED[:]


def _scatter_scratch(idx, weight, b, n_dst, m_src, r):
    """Workspace of the scatter-add gradients -> (scratch, nbytes, mode bits, key).
    0 bytes when the shape is not covered: the entry point then runs the plain
    kernels (zero-filling first, since OVERWRITE is always requested).  `key` is
    not None when the caller has to publish the list it is about to build
    (`_scatter_commit`, after the launch succeeded)."""
    nbytes = scatter_scratch_bytes(b, n_dst, m_src, r)
    if not nbytes:
        return None, 0, SCATTER_OVERWRITE, None
    if not SCATTER_CACHE:
        return torch.empty(nbytes, dtype=torch.uint8, device=idx.device), nbytes, SCATTER_OVERWRITE, None
    key = (idx.data_ptr(), idx._version, tuple(idx.shape), n_dst, idx.device,
           None if weight is None else (weight.data_ptr(), weight._version))
    with _TRANSPOSED_LOCK:
        for i, (k, _, _, scratch) in enumerate(_TRANSPOSED):
            if k == key:
                _TRANSPOSED.append(_TRANSPOSED.pop(i))
                return scratch, nbytes, SCATTER_OVERWRITE | SCATTER_INDEX_READY, None
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=idx.device)
    return scratch, nbytes, SCATTER_OVERWRITE, key


def _scatter_commit(key, idx, weight, scratch):
    """Publish the inverted list the call just built (no-op for key None)."""
    if key is None:
        return
    with _TRANSPOSED_LOCK:
        if any(k == key for k, _, _, _ in _TRANSPOSED):   # another thread built the same list
            return
        _TRANSPOSED.append((key, idx, weight, scratch))
        if len(_TRANSPOSED) > _TRANSPOSED_CAP:
            _TRANSPOSED.pop(0)


# ------------------------------------------------------------------ sampling
FPS_SORTED_MIN_N = 8193      # clouds of at least this many points (<= 16384) take the Morton-sorted kernel (csrc/fps.hip)


class FurthestPointSampling(Function):
    """D-FPS: start at point 0, repeatedly add the point farthest from the
    chosen set.  (B, N, 3) float32, num_points -> (B, num_points) int32."""

    @staticmethod
    def forward(ctx, points_xyz, num_points):
        _need_contiguous(points_xyz)
        B, N = points_xyz.shape[:2]
        out = _new(points_xyz, B, num_points, dtype=torch.int32, zero=True)
        scratch = _new(points_xyz, B, N).fill_(1e10)      # running min-distances
        if FPS_SORTED_MIN_N <= N <= 16384 and num_points > 1:
            # the largest clouds: Morton-sorted copy + wave-level skipping of the distance update (same indices)
            nbytes = fps_scratch_bytes(B, N)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=points_xyz.device)
            call("mvp_furthest_point_sampling_sorted", points_xyz.device, B, N, num_points, points_xyz, scratch, out,
                 ws, nbytes)
        else:
            call("mvp_furthest_point_sampling", points_xyz.device, B, N, num_points, points_xyz, scratch, out)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(xyz, a=None):
        return None, None


class FurthestPointSamplingWithDist(Function):
    """F-FPS on a precomputed (B, N, N) distance matrix -> (B, num_points) int32."""

    @staticmethod
    def forward(ctx, points_dist, num_points):
        _need_contiguous(points_dist)
        B, N, _ = points_dist.shape
        out = _new(points_dist, B, num_points, dtype=torch.int32, zero=True)
        scratch = _new(points_dist, B, N).fill_(1e10)
        call("mvp_furthest_point_sampling_with_dist", points_dist.device, B, N, num_points, points_dist,
             scratch, out)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(xyz, a=None):
        return None, None


# ------------------------------------------------------------------- queries
class BallQuery(Function):
    """For every centre: the first `sample_num` points, in index order, with
    d2 == 0 or min_radius^2 <= d2 < max_radius^2; spare slots repeat the first
    hit, centres without a hit return zeros.
    (min_radius, max_radius, sample_num, xyz (B,N,3), center_xyz (B,M,3)) ->
    (B, M, sample_num) int32."""

    @staticmethod
    def forward(ctx, min_radius, max_radius, sample_num, xyz, center_xyz):
        _need_contiguous(center_xyz, xyz)
        assert min_radius < max_radius
        B, N, _ = xyz.shape
        M = center_xyz.shape[1]
        idx = _new(xyz, B, M, sample_num, dtype=torch.int32, zero=True)
        call("mvp_ball_query", xyz.device, B, N, M, min_radius, max_radius, sample_num, center_xyz, xyz, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None, None


class KNN(Function):
    """k nearest points of every centre (heap order of the reference: ascending
    squared distance), k <= 100.
    (k, xyz, center_xyz=None, transposed=False) -> (B, k, npoint) int32;
    with transposed=True the inputs are (B, 3, N) / (B, 3, npoint)."""

    @staticmethod
    def forward(ctx, k, xyz, center_xyz=None, transposed=False):
        assert k > 0
        center_xyz = xyz if center_xyz is None else center_xyz
        if transposed:
            xyz = xyz.transpose(2, 1).contiguous()
            center_xyz = center_xyz.transpose(2, 1).contiguous()
        _need_contiguous(xyz, center_xyz)
        assert center_xyz.device == xyz.device, 'center_xyz and xyz should be put on the same device'
        B, npoint, _ = center_xyz.shape
        idx = _new(xyz, B, npoint, k, dtype=torch.int32, zero=True)
        dist2 = _new(xyz, B, npoint, k, zero=True)
        n = xyz.shape[1]
        if ((n >= 4096 and npoint >= 1024) or (n >= 2048 and npoint >= 2048)) and k <= 32:
            # large clouds: Morton-sorted, pruned, the same bits (csrc/pn2_query.hip: knn_sorted_kernel)
            nbytes = knn_scratch_bytes(B, n, npoint)
            scratch = torch.empty(nbytes, dtype=torch.uint8, device=xyz.device)
            call("mvp_knn_sorted", xyz.device, B, n, npoint, k, xyz, center_xyz, idx, dist2, scratch, nbytes)
        else:
            call("mvp_knn", xyz.device, B, n, npoint, k, xyz, center_xyz, idx, dist2)
        idx = idx.transpose(2, 1).contiguous()            # (B, k, npoint)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


class ThreeNN(Function):
    """Three nearest `source` points of every `target` point.
    (target (B,N,3), source (B,M,3)) -> (L2 distance (B,N,3) -- NOT squared --,
    indices (B,N,3) int32)."""

    @staticmethod
    def forward(ctx, target, source):
        _need_contiguous(target, source)
        B, N, _ = target.shape
        dist2 = _new(target, B, N, 3)
        idx = _new(target, B, N, 3, dtype=torch.int32)
        call("mvp_three_nn", target.device, B, N, source.shape[1], target, source, dist2, idx)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


# ------------------------------------------------- interpolate / gather / group
class ThreeInterpolate(Function):
    """out[b,c,n] = sum_j weight[b,n,j] * features[b,c,indices[b,n,j]].
    (features (B,C,M), indices (B,n,3) int32, weight (B,n,3)) -> (B,C,n);
    differentiable w.r.t. features."""

    @staticmethod
    def forward(ctx, features, indices, weight):
        _need_contiguous(features, indices, weight)
        B, c, m = features.shape
        n = indices.shape[1]
        ctx.three_interpolate_for_backward = (indices, weight, m)
        out = _new(features, B, c, n)
        call("mvp_three_interpolate", features.device, B, c, m, n, features, indices, weight, out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, m = ctx.three_interpolate_for_backward
        B, c, n = grad_out.shape
        grad_features = _new(grad_out, B, c, m)          # written, not accumulated into (OVERWRITE)
        scratch, nbytes, mode, key = _scatter_scratch(idx, weight, B, m, n, 3)
        call("mvp_three_interpolate_grad_ws", grad_out.device, B, c, n, m, grad_out.data.contiguous(), idx, weight,
             grad_features, scratch, nbytes, mode)
        _scatter_commit(key, idx, weight, scratch)
        return grad_features, None, None


class GatherPoints(Function):
    """out[b,c,m] = features[b,c,indices[b,m]].
    (features (B,C,N), indices (B,M) int32) -> (B,C,M); differentiable w.r.t.
    features (scatter-add)."""

    @staticmethod
    def forward(ctx, features, indices):
        _need_contiguous(features, indices)
        B, npoint = indices.shape
        _, C, N = features.shape
        out = _new(features, B, C, npoint)
        call("mvp_gather_points", features.device, B, C, N, npoint, features, indices, out)
        ctx.for_backwards = (indices, C, N)
        ctx.mark_non_differentiable(indices)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        B, npoint = idx.shape
        grad_features = _new(grad_out, B, C, N)
        scratch, nbytes, mode, key = _scatter_scratch(idx, None, B, N, npoint, 1)
        call("mvp_gather_points_grad_ws", grad_out.device, B, C, N, npoint, grad_out.data.contiguous(), idx,
             grad_features, scratch, nbytes, mode)
        _scatter_commit(key, idx, None, scratch)
        return grad_features, None


class GroupingOperation(Function):
    """out[b,c,p,s] = features[b,c,indices[b,p,s]].
    (features (B,C,N), indices (B,npoint,nsample) int32) -> (B,C,npoint,nsample);
    differentiable w.r.t. features (scatter-add)."""

    @staticmethod
    def forward(ctx, features, indices):
        _need_contiguous(features, indices)
        B, npoint, nsample = indices.shape
        _, C, N = features.shape
        out = _new(features, B, C, npoint, nsample)
        call("mvp_group_points", features.device, B, C, N, npoint, nsample, features, indices, out)
        ctx.for_backwards = (indices, N)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, N = ctx.for_backwards
        B, C, npoint, nsample = grad_out.shape
        grad_features = _new(grad_out, B, C, N)
        scratch, nbytes, mode, key = _scatter_scratch(idx, None, B, N, npoint * nsample, 1)
        call("mvp_group_points_grad_ws", grad_out.device, B, C, N, npoint, nsample, grad_out.data.contiguous(), idx,
             grad_features, scratch, nbytes, mode)
        _scatter_commit(key, idx, None, scratch)
        return grad_features, None


class GatherMax(Function):
    """out[b,c,p] = max_j features[b,c, indices[b,p,j]] -- the gather_points + torch.max pair of
    edge_preserve_sampling fused (mvp_gather_max: the (B,C,P,k) neighbour tensor is never
    written).  (features (B,C,N) float32, indices (B,P,k) int32) -> (B,C,P); differentiable w.r.t.
    features: the gradient goes to the first maximal neighbour, torch.max's rule.  Not part of the
    reference's operator set (row N2 of the widening plan)."""

    @staticmethod
    def forward(ctx, features, indices):
        _need_contiguous(features, indices)
        B, C, N = features.shape
        _, P, K = indices.shape
        out = _new(features, B, C, P)
        arg = _new(features, B, C, P, dtype=torch.int32)
        call("mvp_gather_max", features.device, B, C, N, P, K, features, indices, out, arg)
        ctx.for_backwards = (arg, N)
        ctx.mark_non_differentiable(indices)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        arg, N = ctx.for_backwards
        B, C, P = grad_out.shape
        grad_features = _new(grad_out, B, C, N)          # written, not accumulated into
        call("mvp_gather_max_grad", grad_out.device, B, C, N, P, grad_out.data.contiguous(), arg, grad_features, 1)
        return grad_features, None


GATHER_MAX_LIMIT = 24576      # floats of one feature row the fused kernel can stage (96 KiB of LDS)


def gather_max(features, indices):
    """Fused neighbour max-pool (see GatherMax); rows too long for the LDS staging buffer take the
    gather + reduce route through the gather operator."""
    if features.size(2) > GATHER_MAX_LIMIT:
        B, C, _ = features.shape
        _, P, K = indices.shape
        flat = indices.transpose(1, 2).contiguous().view(B, K * P)
        return GatherPoints.apply(features, flat).view(B, C, K, P).max(dim=2)[0]
    return GatherMax.apply(features, indices)


def gram_topk(dot, sq, k):
    """Feature-space neighbours: dot (B,N,N) = x^T x, sq (B,N) = |x_i|^2 ->
    idx (B,N,k) int32, per row the k largest of (-sq[j] + 2 dot[i][j]) - sq[i]
    in descending order (self first) -- the selection model_utils.knn makes with
    torch.topk on the materialised negative-distance matrix, in one scan.  Not
    part of the reference's operator set (row N1 of the widening plan)."""
    _need_contiguous(dot, sq)
    B, N = sq.shape
    idx = _new(dot, B, N, k, dtype=torch.int32)
    call("mvp_topk_gram", dot.device, B, N, k, dot, sq, idx)
    return idx


class ShareWeightedSum(Function):
    """out[b, s*Cw + m, n] = sum_k w[b, m, k, n] * v[b, s*Cw + m, k, n]:
    (w (B,Cw,k,N), v (B,share*Cw,k,N)) -> (B, share*Cw, N), differentiable in
    both.  The neighbourhood aggregation of VRCNet's SA_module (vrcnet.py:52-55:
    `w.repeat(1, share, 1, 1)`, product, sum over k) as one streaming pass each
    way.  Not part of the reference's operator set."""

    SHARES = (1, 2, 4, 8, 16)

    @staticmethod
    def forward(ctx, w, v):
        _need_contiguous(w, v)
        B, Cw, k, N = w.shape
        C = v.shape[1]
        share = C // max(Cw, 1)
        assert v.shape == (B, share * Cw, k, N) and share in ShareWeightedSum.SHARES
        out = _new(v, B, C, N)
        call("mvp_share_weighted_sum", v.device, B, share, Cw, k, N, w, v, out)
        ctx.save_for_backward(w, v)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        w, v = ctx.saved_tensors
        B, Cw, k, N = w.shape
        share = v.shape[1] // max(Cw, 1)
        grad_w, grad_v = torch.empty_like(w), torch.empty_like(v)
        if k == 0:
            return grad_w, grad_v
        call("mvp_share_weighted_sum_grad", v.device, B, share, Cw, k, N, w, v, grad_out.data.contiguous(),
             grad_w, grad_v)
        return grad_w, grad_v


class ShareGatherSum(Function):
    """out[b, s*Cw + m, p] = sum_k w[b, m, k, p] * v[b, s*Cw + m, idx[b, k, p]]: ShareWeightedSum with the gather of the
    neighbours' values (the grouping operator) fused in.  The FORWARD never forms the (B, share*Cw, k, N) tensor of gathered
    values (written by the grouping kernel, read by the weighted sum and kept for its gradient in the two-step route); the
    BACKWARD gathers again, emits grad_w and -- when v needs a gradient -- the gathered values' gradient as ONE
    (B, share*Cw, k, N) temporary, which the grouping operator's scatter (inverted index, no atomics) sums into grad_v.
    (w (B,Cw,k,N), v (B,share*Cw,Nsrc), idx (B,k,N) int32) -> (B, share*Cw, N); differentiable in w and v; bit-identical
    to share_weighted_sum(w, grouping_operation(v, idx))."""

    LDS_BYTES = 96 * 1024

    @staticmethod
    def covers(share, n_src):
        return share in ShareWeightedSum.SHARES and share * n_src * 4 <= ShareGatherSum.LDS_BYTES

    @staticmethod
    def forward(ctx, w, v, idx):
        _need_contiguous(w, v, idx)
        B, Cw, k, N = w.shape
        C, n_src = v.shape[1], v.shape[2]
        share = C // max(Cw, 1)
        assert v.shape[0] == B and share * Cw == C and idx.shape == (B, k, N) and ShareGatherSum.covers(share, n_src)
        out = _new(v, B, C, N)
        call("mvp_share_gather_sum", v.device, B, share, Cw, k, n_src, N, w, v, idx, out)
        ctx.save_for_backward(w, v, idx)
        ctx.mark_non_differentiable(idx)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        w, v, idx = ctx.saved_tensors
        B, Cw, k, N = w.shape
        C, n_src = v.shape[1], v.shape[2]
        share = C // max(Cw, 1)
        grad_w = torch.empty_like(w)
        grad_v = _new(v, B, C, n_src)
        if k == 0:
            return grad_w, grad_v.zero_(), None
        grad_vals = _new(v, B, C, k, N)
        call("mvp_share_gather_sum_grad", v.device, B, share, Cw, k, n_src, N, w, v, idx, grad_out.data.contiguous(),
             grad_w, grad_vals)
        if not ctx.needs_input_grad[1]:          # (frozen values: no scatter pass)
            return grad_w, None, None
        scratch, nbytes, mode, key = _scatter_scratch(idx, None, B, n_src, k * N, 1)
        call("mvp_group_points_grad_ws", v.device, B, C, n_src, k, N, grad_vals, idx, grad_v, scratch, nbytes, mode)
        _scatter_commit(key, idx, None, scratch)
        return grad_w, grad_v, None


furthest_point_sample = FurthestPointSampling.apply
share_weighted_sum = ShareWeightedSum.apply
share_gather_sum = ShareGatherSum.apply
furthest_point_sample_with_dist = FurthestPointSamplingWithDist.apply
ball_query = BallQuery.apply
knn = KNN.apply
three_nn = ThreeNN.apply
three_interpolate = ThreeInterpolate.apply
gather_points = GatherPoints.apply
grouping_operation = GroupingOperation.apply
