"""Gather points -- same surface as the reference's
utils/mm3d_pn2/ops/gather_points/gather_points.py:7-52, backed by
mvp_gather_points / mvp_gather_points_grad."""
import torch
from torch.autograd import Function

from ...._lib import call


class GatherPoints(Function):
    """out[b, c, m] = features[b, c, indices[b, m]]"""

    @staticmethod
    def forward(ctx, features: torch.Tensor,
                indices: torch.Tensor) -> torch.Tensor:
        """
        Args:
            features (Tensor): (B, C, N) features to gather.
            indices (Tensor): (B, M) int32, M = number of points.

        Returns:
            Tensor: (B, C, M)
        """
        assert features.is_contiguous()
        assert indices.is_contiguous()

        B, npoint = indices.size()
        _, C, N = features.size()
        output = torch.empty(B, C, npoint, dtype=torch.float32,
                             device=features.device)
        call("mvp_gather_points", features.device, B, C, N, npoint, features,
             indices, output)
        ctx.for_backwards = (indices, C, N)
        ctx.mark_non_differentiable(indices)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        B, npoint = idx.size()

        grad_features = torch.zeros(B, C, N, dtype=torch.float32,
                                    device=grad_out.device)
        grad_out_data = grad_out.data.contiguous()
        call("mvp_gather_points_grad", grad_out.device, B, C, N, npoint,
             grad_out_data, idx, grad_features)
        return grad_features, None


gather_points = GatherPoints.apply
