"""Ball query -- same surface as the reference's
utils/mm3d_pn2/ops/ball_query/ball_query.py:7-47, backed by mvp_ball_query."""
import torch
from torch.autograd import Function

from ...._lib import call


class BallQuery(Function):
    """Find, for every centre, the first ``sample_num`` points (in index
    order) whose squared distance d2 satisfies
    ``d2 == 0 or min_radius**2 <= d2 < max_radius**2``; unused slots repeat
    the first hit; centres without any hit return zeros."""

    @staticmethod
    def forward(ctx, min_radius: float, max_radius: float, sample_num: int,
                xyz: torch.Tensor, center_xyz: torch.Tensor) -> torch.Tensor:
        """
        Args:
            min_radius (float): minimum radius of the balls.
            max_radius (float): maximum radius of the balls.
            sample_num (int): maximum number of features in the balls.
            xyz (Tensor): (B, N, 3) xyz coordinates of the features.
            center_xyz (Tensor): (B, npoint, 3) centers of the ball query.

        Returns:
            Tensor: (B, npoint, nsample) int32 indices.
        """
        assert center_xyz.is_contiguous()
        assert xyz.is_contiguous()
        assert min_radius < max_radius

        B, N, _ = xyz.size()
        npoint = center_xyz.size(1)
        idx = torch.zeros(B, npoint, sample_num, dtype=torch.int32,
                          device=xyz.device)
        call("mvp_ball_query", xyz.device, B, N, npoint, min_radius,
             max_radius, sample_num, center_xyz, xyz, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None, None


ball_query = BallQuery.apply
