"""Grouping -- same surface as the reference's
utils/mm3d_pn2/ops/group_points/group_points.py (QueryAndGroup :11-122,
GroupAll :125-163, GroupingOperation :166-221), backed by mvp_group_points /
mvp_group_points_grad."""
from typing import Tuple

import torch
from torch import nn as nn
from torch.autograd import Function

from ...._lib import call
from ..ball_query import ball_query
from ..knn import knn


class QueryAndGroup(nn.Module):
    """Neighbourhood query (ball query, or kNN when ``max_radius`` is None)
    followed by grouping of centred xyz and features.

    Args:
        max_radius (float | None): maximum ball radius; None selects kNN.
        sample_num (int): neighbours per centre.
        min_radius (float): minimum ball radius. Default: 0.
        use_xyz (bool): concatenate grouped xyz. Default: True.
        return_grouped_xyz (bool): also return grouped xyz. Default: False.
        normalize_xyz (bool): divide grouped xyz by max_radius. Default: False.
        uniform_sample (bool): resample duplicates uniformly. Default: False.
        return_unique_cnt (bool): also return the count of unique samples
            (needs uniform_sample). Default: False.
    """

    def __init__(self,
                 max_radius,
                 sample_num,
                 min_radius=0,
                 use_xyz=True,
                 return_grouped_xyz=False,
                 normalize_xyz=False,
                 uniform_sample=False,
                 return_unique_cnt=False):
        super(QueryAndGroup, self).__init__()
        self.max_radius = max_radius
        self.min_radius = min_radius
        self.sample_num = sample_num
        self.use_xyz = use_xyz
        self.return_grouped_xyz = return_grouped_xyz
        self.normalize_xyz = normalize_xyz
        self.uniform_sample = uniform_sample
        self.return_unique_cnt = return_unique_cnt
        if self.return_unique_cnt:
            assert self.uniform_sample, \
                'uniform_sample should be True when ' \
                'returning the count of unique samples'
        if self.max_radius is None:
            assert not self.normalize_xyz, \
                'can not normalize grouped xyz when max_radius is None'

    def forward(self, points_xyz, center_xyz, features=None):
        """
        Args:
            points_xyz (Tensor): (B, N, 3) xyz coordinates of the features.
            center_xyz (Tensor): (B, npoint, 3) centroids.
            features (Tensor): (B, C, N) descriptors of the features.

        Return:
            Tensor: (B, 3 + C, npoint, sample_num) grouped feature.
        """
        if self.max_radius is None:
            idx = knn(self.sample_num, points_xyz, center_xyz, False)
            idx = idx.transpose(1, 2).contiguous()
        else:
            idx = ball_query(self.min_radius, self.max_radius, self.sample_num,
                             points_xyz, center_xyz)

        if self.uniform_sample:
            unique_cnt = torch.zeros((idx.shape[0], idx.shape[1]))
            for i_batch in range(idx.shape[0]):
                for i_region in range(idx.shape[1]):
                    unique_ind = torch.unique(idx[i_batch, i_region, :])
                    num_unique = unique_ind.shape[0]
                    unique_cnt[i_batch, i_region] = num_unique
                    sample_ind = torch.randint(
                        0, num_unique, (self.sample_num - num_unique, ),
                        dtype=torch.long, device=unique_ind.device)
                    all_ind = torch.cat((unique_ind, unique_ind[sample_ind]))
                    idx[i_batch, i_region, :] = all_ind

        xyz_trans = points_xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx)  # (B, 3, npoint, S)
        grouped_xyz = grouped_xyz - center_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped_xyz = grouped_xyz / self.max_radius

        if features is not None:
            grouped_features = grouping_operation(features, idx)
            if self.use_xyz:
                new_features = torch.cat([grouped_xyz, grouped_features], dim=1)
            else:
                new_features = grouped_features
        else:
            assert (self.use_xyz
                    ), 'Cannot have not features and not use xyz as a feature!'
            new_features = grouped_xyz

        ret = [new_features]
        if self.return_grouped_xyz:
            ret.append(grouped_xyz)
        if self.return_unique_cnt:
            ret.append(unique_cnt)
        return ret[0] if len(ret) == 1 else tuple(ret)


class GroupAll(nn.Module):
    """Group every point into one region.

    Args:
        use_xyz (bool): concatenate xyz in front of the features.
    """

    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self,
                xyz: torch.Tensor,
                new_xyz: torch.Tensor,
                features: torch.Tensor = None):
        """
        Args:
            xyz (Tensor): (B, N, 3) xyz coordinates of the features.
            new_xyz (Tensor): ignored.
            features (Tensor): (B, C, N) features to group.

        Return:
            Tensor: (B, C + 3, 1, N) grouped feature.
        """
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped_features = features.unsqueeze(2)
        if self.use_xyz:
            return torch.cat([grouped_xyz, grouped_features], dim=1)
        return grouped_features


class GroupingOperation(Function):
    """out[b, c, p, s] = features[b, c, indices[b, p, s]]"""

    @staticmethod
    def forward(ctx, features: torch.Tensor,
                indices: torch.Tensor) -> torch.Tensor:
        """
        Args:
            features (Tensor): (B, C, N) tensor of features to group.
            indices (Tensor): (B, npoint, nsample) int32 indices.

        Returns:
            Tensor: (B, C, npoint, nsample) grouped features.
        """
        assert features.is_contiguous()
        assert indices.is_contiguous()

        B, nfeatures, nsample = indices.size()
        _, C, N = features.size()
        output = torch.empty(B, C, nfeatures, nsample, dtype=torch.float32,
                             device=features.device)
        call("mvp_group_points", features.device, B, C, N, nfeatures, nsample,
             features, indices, output)
        ctx.for_backwards = (indices, N)
        return output

    @staticmethod
    def backward(ctx,
                 grad_out: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """
        Args:
            grad_out (Tensor): (B, C, npoint, nsample) gradient of the output.

        Returns:
            Tensor: (B, C, N) gradient of the features.
        """
        idx, N = ctx.for_backwards

        B, C, npoint, nsample = grad_out.size()
        grad_features = torch.zeros(B, C, N, dtype=torch.float32,
                                    device=grad_out.device)
        grad_out_data = grad_out.data.contiguous()
        call("mvp_group_points_grad", grad_out.device, B, C, N, npoint,
             nsample, grad_out_data, idx, grad_features)
        return grad_features, None


grouping_operation = GroupingOperation.apply
