"""Counterpart of utils/mm3d_pn2/ops/__init__.py:1-47 (hot-path names)."""
from .ball_query import ball_query
from .furthest_point_sample import (Points_Sampler, furthest_point_sample,
                                    furthest_point_sample_with_dist)
from .gather_points import gather_points
from .group_points import (GroupAll, QueryAndGroup, group_points,
                           grouping_operation)
from .interpolate import three_interpolate, three_nn
from .knn import knn
from .norm import NaiveSyncBatchNorm1d, NaiveSyncBatchNorm2d

__all__ = [
    'NaiveSyncBatchNorm1d', 'NaiveSyncBatchNorm2d',
    'ball_query', 'knn', 'furthest_point_sample',
    'furthest_point_sample_with_dist', 'three_interpolate', 'three_nn',
    'gather_points', 'grouping_operation', 'group_points', 'GroupAll',
    'QueryAndGroup', 'Points_Sampler',
]

# Detection-only re-exports of the reference (ops/__init__.py:1-3) come from
# the un-vendored mmcv; they are not part of this op layer.
try:  # pragma: no cover - mmcv is not installed in the build image
    from mmcv.ops import (RoIAlign, SigmoidFocalLoss, get_compiler_version,
                          get_compiling_cuda_version, nms, roi_align,
                          sigmoid_focal_loss)
    __all__ += ['nms', 'RoIAlign', 'roi_align', 'get_compiler_version',
                'get_compiling_cuda_version', 'sigmoid_focal_loss',
                'SigmoidFocalLoss']
except ImportError:
    pass
