"""Furthest point sampling -- same surface as the reference's
utils/mm3d_pn2/ops/furthest_point_sample/furthest_point_sample.py:7-78,
backed by mvp_furthest_point_sampling[_with_dist]."""
import torch
from torch.autograd import Function

from ...._lib import call


class FurthestPointSampling(Function):
    """Iterative furthest point sampling on xyz (D-FPS)."""

    @staticmethod
    def forward(ctx, points_xyz: torch.Tensor,
                num_points: int) -> torch.Tensor:
        """
        Args:
            points_xyz (Tensor): (B, N, 3) where N > num_points.
            num_points (int): Number of points in the sampled set.

        Returns:
             Tensor: (B, num_points) int32 indices of the sampled points.
        """
        assert points_xyz.is_contiguous()

        B, N = points_xyz.size()[:2]
        output = torch.zeros(B, num_points, dtype=torch.int32,
                             device=points_xyz.device)
        temp = torch.empty(B, N, dtype=torch.float32,
                           device=points_xyz.device).fill_(1e10)
        call("mvp_furthest_point_sampling", points_xyz.device, B, N,
             num_points, points_xyz, temp, output)
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(xyz, a=None):
        return None, None


class FurthestPointSamplingWithDist(Function):
    """Furthest point sampling on a precomputed (B, N, N) distance matrix
    (F-FPS)."""

    @staticmethod
    def forward(ctx, points_dist: torch.Tensor,
                num_points: int) -> torch.Tensor:
        """
        Args:
            points_dist (Tensor): (B, N, N) Distance between each point pair.
            num_points (int): Number of points in the sampled set.

        Returns:
             Tensor: (B, num_points) int32 indices of the sampled points.
        """
        assert points_dist.is_contiguous()

        B, N, _ = points_dist.size()
        output = points_dist.new_zeros([B, num_points], dtype=torch.int32)
        temp = points_dist.new_zeros([B, N]).fill_(1e10)
        call("mvp_furthest_point_sampling_with_dist", points_dist.device, B, N,
             num_points, points_dist, temp, output)
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(xyz, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply
furthest_point_sample_with_dist = FurthestPointSamplingWithDist.apply
