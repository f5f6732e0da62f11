"""Range-sliced point sampler facade -- counterpart of the reference's
utils/mm3d_pn2/ops/furthest_point_sample/points_sampler.py:34-158.  The
reference decorates ``forward`` with mmcv's ``force_fp32`` (:2,65); mmcv is
not a dependency here, inputs are cast to float32 explicitly instead."""
from typing import List

import torch
from torch import nn as nn

from .furthest_point_sample import (furthest_point_sample,
                                    furthest_point_sample_with_dist)
from .utils import calc_square_dist


def get_sampler_type(sampler_type):
    """Map "D-FPS" / "F-FPS" / "FS" to the sampler class."""
    table = {'D-FPS': DFPS_Sampler, 'F-FPS': FFPS_Sampler, 'FS': FS_Sampler}
    if sampler_type not in table:
        raise ValueError('Only "sampler_type" of "D-FPS", "F-FPS", or "FS"'
                         f' are supported, got {sampler_type}')
    return table[sampler_type]


class Points_Sampler(nn.Module):
    """Apply one FPS flavour per index range of the input cloud.

    Args:
        num_point (list[int]): Number of sample points per range.
        fps_mod_list (list[str]): 'F-FPS' (feature distance), 'D-FPS'
            (Euclidean distance) or 'FS' (both). Default: ['D-FPS'].
        fps_sample_range_list (list[int]): end index of each range, -1 = to
            the end. Default: [-1].
    """

    def __init__(self,
                 num_point: List[int],
                 fps_mod_list: List[str] = ['D-FPS'],
                 fps_sample_range_list: List[int] = [-1]):
        super(Points_Sampler, self).__init__()
        assert len(num_point) == len(fps_mod_list) == len(
            fps_sample_range_list)
        self.num_point = num_point
        self.fps_sample_range_list = fps_sample_range_list
        self.samplers = nn.ModuleList(
            [get_sampler_type(mod)() for mod in fps_mod_list])
        self.fp16_enabled = False

    def forward(self, points_xyz, features):
        """
        Args:
            points_xyz (Tensor): (B, N, 3) xyz coordinates of the features.
            features (Tensor): (B, C, N) descriptors of the features.

        Return:
            Tensor: (B, sum(num_point)) indices of sampled points.
        """
        points_xyz = points_xyz.float()
        features = features.float() if features is not None else None
        indices = []
        last_fps_end_index = 0
        for fps_sample_range, sampler, npoint in zip(
                self.fps_sample_range_list, self.samplers, self.num_point):
            assert fps_sample_range < points_xyz.shape[1]
            end = None if fps_sample_range == -1 else fps_sample_range
            sample_points_xyz = points_xyz[:, last_fps_end_index:end]
            sample_features = None if features is None else \
                features[:, :, last_fps_end_index:end]
            fps_idx = sampler(sample_points_xyz.contiguous(), sample_features,
                              npoint)
            indices.append(fps_idx + last_fps_end_index)
            last_fps_end_index += fps_sample_range
        return torch.cat(indices, dim=1)


class DFPS_Sampler(nn.Module):
    """FPS on Euclidean distances."""

    def forward(self, points, features, npoint):
        return furthest_point_sample(points.contiguous(), npoint)


def _feature_dist(points, features):
    feats = torch.cat([points, features.transpose(1, 2)], dim=2)
    return calc_square_dist(feats, feats, norm=False).contiguous()


class FFPS_Sampler(nn.Module):
    """FPS on feature distances."""

    def forward(self, points, features, npoint):
        assert features is not None, \
            'feature input to FFPS_Sampler should not be None'
        return furthest_point_sample_with_dist(
            _feature_dist(points, features), npoint)


class FS_Sampler(nn.Module):
    """F-FPS and D-FPS side by side."""

    def forward(self, points, features, npoint):
        assert features is not None, \
            'feature input to FS_Sampler should not be None'
        fps_idx_ffps = furthest_point_sample_with_dist(
            _feature_dist(points, features), npoint)
        fps_idx_dfps = furthest_point_sample(points.contiguous(), npoint)
        return torch.cat([fps_idx_ffps, fps_idx_dfps], dim=1)
