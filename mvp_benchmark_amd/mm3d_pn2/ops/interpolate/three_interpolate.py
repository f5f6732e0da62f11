"""Three-point weighted interpolation -- same surface as the reference's
utils/mm3d_pn2/ops/interpolate/three_interpolate.py:8-63, backed by
mvp_three_interpolate / mvp_three_interpolate_grad."""
from typing import Tuple

import torch
from torch.autograd import Function

from ...._lib import call


class ThreeInterpolate(Function):

    @staticmethod
    def forward(ctx, features: torch.Tensor, indices: torch.Tensor,
                weight: torch.Tensor) -> torch.Tensor:
        """
        Args:
            features (Tensor): (B, C, M) features to interpolate from.
            indices (Tensor): (B, n, 3) int32 indices of the 3 neighbours.
            weight (Tensor): (B, n, 3) interpolation weights.

        Returns:
            Tensor: (B, C, n) interpolated features.
        """
        assert features.is_contiguous()
        assert indices.is_contiguous()
        assert weight.is_contiguous()

        B, c, m = features.size()
        n = indices.size(1)
        ctx.three_interpolate_for_backward = (indices, weight, m)
        output = torch.empty(B, c, n, dtype=torch.float32,
                             device=features.device)
        call("mvp_three_interpolate", features.device, B, c, m, n, features,
             indices, weight, output)
        return output

    @staticmethod
    def backward(
        ctx, grad_out: torch.Tensor
    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """
        Args:
            grad_out (Tensor): (B, C, n) gradient of the output.

        Returns:
            Tensor: (B, C, M) gradient of the features.
        """
        idx, weight, m = ctx.three_interpolate_for_backward
        B, c, n = grad_out.size()

        grad_features = torch.zeros(B, c, m, dtype=torch.float32,
                                    device=grad_out.device)
        grad_out_data = grad_out.data.contiguous()
        call("mvp_three_interpolate_grad", grad_out.device, B, c, n, m,
             grad_out_data, idx, weight, grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply
