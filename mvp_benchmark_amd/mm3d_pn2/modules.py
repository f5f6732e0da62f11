"""nn.Module compositions of the operators -- counterparts of the reference's
QueryAndGroup / GroupAll (utils/mm3d_pn2/ops/group_points/group_points.py:
11-163), Points_Sampler and its D-FPS / F-FPS / FS samplers
(ops/furthest_point_sample/points_sampler.py:34-158) and calc_square_dist
(ops/furthest_point_sample/utils.py:4-31).  Same constructor arguments and
return values; the reference's `mmcv.runner.force_fp32` decorator is replaced
by an explicit float32 cast (mmcv is not a dependency here)."""
from typing import List

import torch
from torch import nn as nn

from .functional import (ball_query, furthest_point_sample, furthest_point_sample_with_dist,
                         grouping_operation, knn)


def calc_square_dist(point_feat_a, point_feat_b, norm=True):
    """(B,N,C), (B,M,C) -> (B,N,M) squared feature distance |a|^2 + |b|^2 - 2ab;
    with norm=True its square root divided by C."""
    sq_a = point_feat_a.pow(2).sum(dim=-1, keepdim=True)       # (B, N, 1)
    sq_b = point_feat_b.pow(2).sum(dim=-1).unsqueeze(1)        # (B, 1, M)
    dist = sq_a + sq_b - 2 * torch.matmul(point_feat_a, point_feat_b.transpose(1, 2))
    return torch.sqrt(dist) / point_feat_a.shape[-1] if norm else dist


class QueryAndGroup(nn.Module):
    """Neighbourhood query followed by grouping.

    Args:
        max_radius (float | None): ball radius; None selects kNN.
        sample_num (int): neighbours per centre.
        min_radius (float): inner radius of the ball. Default 0.
        use_xyz (bool): prepend the centred xyz to the grouped features.
        return_grouped_xyz (bool): also return the grouped xyz.
        normalize_xyz (bool): divide the centred xyz by max_radius.
        uniform_sample (bool): replace duplicated neighbours by uniform draws
            from the unique ones.
        return_unique_cnt (bool): also return the unique-neighbour counts
            (requires uniform_sample).
    forward(points_xyz (B,N,3), center_xyz (B,P,3), features (B,C,N) | None)
        -> (B, 3 + C, P, sample_num) [, grouped_xyz] [, unique_cnt]
    """

    def __init__(self, max_radius, sample_num, min_radius=0, use_xyz=True, return_grouped_xyz=False,
                 normalize_xyz=False, uniform_sample=False, return_unique_cnt=False):
        super(QueryAndGroup, self).__init__()
        if return_unique_cnt:
            assert uniform_sample, 'uniform_sample should be True when returning the count of unique samples'
        if max_radius is None:
            assert not normalize_xyz, 'can not normalize grouped xyz when max_radius is None'
        self.max_radius, self.min_radius, self.sample_num = max_radius, min_radius, sample_num
        self.use_xyz, self.return_grouped_xyz = use_xyz, return_grouped_xyz
        self.normalize_xyz, self.uniform_sample = normalize_xyz, uniform_sample
        self.return_unique_cnt = return_unique_cnt

    def _neighbours(self, points_xyz, center_xyz):
        if self.max_radius is None:
            return knn(self.sample_num, points_xyz, center_xyz, False).transpose(1, 2).contiguous()
        return ball_query(self.min_radius, self.max_radius, self.sample_num, points_xyz, center_xyz)

    def _resample_unique(self, idx):
        """In place: every row keeps its unique indices and fills the rest
        with uniform draws from them; returns the unique counts."""
        counts = torch.zeros(idx.shape[:2])
        for b in range(idx.shape[0]):
            for p in range(idx.shape[1]):
                uniq = torch.unique(idx[b, p])
                counts[b, p] = uniq.numel()
                # (drawn on the HOST generator, as the reference does -- group_points.py:86-90 calls torch.randint without a
                # device --, so that a seed reproduces the reference's draws on any device)
                extra = torch.randint(0, uniq.numel(), (self.sample_num - uniq.numel(),), dtype=torch.long)
                idx[b, p] = torch.cat((uniq, uniq[extra.to(uniq.device)]))
        return counts

    def forward(self, points_xyz, center_xyz, features=None):
        idx = self._neighbours(points_xyz, center_xyz)                     # (B, P, S) int32
        unique_cnt = self._resample_unique(idx) if self.uniform_sample else None

        grouped_xyz = grouping_operation(points_xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz = grouped_xyz - center_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped_xyz = grouped_xyz / self.max_radius

        if features is None:
            assert self.use_xyz, 'Cannot have not features and not use xyz as a feature!'
            new_features = grouped_xyz
        else:
            grouped = grouping_operation(features, idx)
            new_features = torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped

        outs = [new_features]
        if self.return_grouped_xyz:
            outs.append(grouped_xyz)
        if self.return_unique_cnt:
            outs.append(unique_cnt)
        return outs[0] if len(outs) == 1 else tuple(outs)


class GroupAll(nn.Module):
    """One group holding every point: (B,N,3) [, (B,C,N)] -> (B, 3 + C, 1, N)."""

    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped = features.unsqueeze(2)
        return torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped


# ---------------------------------------------------------------- samplers
class DFPS_Sampler(nn.Module):
    """FPS on Euclidean distances."""

    def forward(self, points, features, npoint):
        return furthest_point_sample(points.contiguous(), npoint)


def _joint_feature_distance(points, features):
    joint = torch.cat([points, features.transpose(1, 2)], dim=2)
    return calc_square_dist(joint, joint, norm=False).contiguous()


class FFPS_Sampler(nn.Module):
    """FPS on the distance of [xyz, features]."""

    def forward(self, points, features, npoint):
        assert features is not None, 'feature input to FFPS_Sampler should not be None'
        return furthest_point_sample_with_dist(_joint_feature_distance(points, features), npoint)


class FS_Sampler(nn.Module):
    """F-FPS and D-FPS results side by side."""

    def forward(self, points, features, npoint):
        assert features is not None, 'feature input to FS_Sampler should not be None'
        by_feature = furthest_point_sample_with_dist(_joint_feature_distance(points, features), npoint)
        by_xyz = furthest_point_sample(points.contiguous(), npoint)
        return torch.cat([by_feature, by_xyz], dim=1)


_SAMPLERS = {'D-FPS': DFPS_Sampler, 'F-FPS': FFPS_Sampler, 'FS': FS_Sampler}


def get_sampler_type(sampler_type):
    if sampler_type not in _SAMPLERS:
        raise ValueError('Only "sampler_type" of "D-FPS", "F-FPS", or "FS"'
                         f' are supported, got {sampler_type}')
    return _SAMPLERS[sampler_type]


class Points_Sampler(nn.Module):
    """One FPS flavour per index range of the cloud.

    Args:
        num_point (list[int]): samples per range.
        fps_mod_list (list[str]): 'D-FPS' | 'F-FPS' | 'FS' per range.
        fps_sample_range_list (list[int]): end index of each range (-1 = rest).
    forward(points_xyz (B,N,3), features (B,C,N)) -> (B, sum(num_point)) indices
    """

    def __init__(self, num_point: List[int], fps_mod_list: List[str] = ['D-FPS'],
                 fps_sample_range_list: List[int] = [-1]):
        super(Points_Sampler, self).__init__()
        assert len(num_point) == len(fps_mod_list) == len(fps_sample_range_list)
        self.num_point = num_point
        self.fps_sample_range_list = fps_sample_range_list
        self.samplers = nn.ModuleList([get_sampler_type(mod)() for mod in fps_mod_list])
        self.fp16_enabled = False

    def forward(self, points_xyz, features):
        points_xyz = points_xyz.float()
        features = None if features is None else features.float()
        picked, start = [], 0
        for end, sampler, npoint in zip(self.fps_sample_range_list, self.samplers, self.num_point):
            assert end < points_xyz.shape[1]
            stop = None if end == -1 else end
            part_feat = None if features is None else features[:, :, start:stop]
            picked.append(sampler(points_xyz[:, start:stop].contiguous(), part_feat, npoint) + start)
            start += end
        return torch.cat(picked, dim=1)
