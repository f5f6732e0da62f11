"""Counterpart of the reference's ``utils/mm3d_pn2`` package
(utils/mm3d_pn2/__init__.py:1-19): the PointNet++ set-abstraction operators.

The reference also re-exports detection-only names from the un-vendored
``mmcv.ops`` (nms, RoIAlign, roi_align, SigmoidFocalLoss, sigmoid_focal_loss,
get_compiler_version, get_compiling_cuda_version); those are outside the hot
path and are re-exported only when mmcv is
importable.
"""
from .ops import (NaiveSyncBatchNorm1d, NaiveSyncBatchNorm2d, ball_query, knn,
                  furthest_point_sample, furthest_point_sample_with_dist,
                  three_interpolate, three_nn, gather_points,
                  grouping_operation, group_points, GroupAll, QueryAndGroup,
                  Points_Sampler)
from .ops import __all__ as _ops_all
from . import ops as _ops

for _name in _ops_all:
    globals().setdefault(_name, getattr(_ops, _name))

__all__ = list(_ops_all)
