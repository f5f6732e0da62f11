"""Three nearest neighbours -- same surface as the reference's
utils/mm3d_pn2/ops/interpolate/three_nn.py:8-45, backed by mvp_three_nn."""
from typing import Tuple

import torch
from torch.autograd import Function

from ...._lib import call


class ThreeNN(Function):

    @staticmethod
    def forward(ctx, target: torch.Tensor,
                source: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Top-3 nearest neighbours of every target point in the source set.

        Args:
            target (Tensor): (B, N, 3) points that need neighbours.
            source (Tensor): (B, M, 3) points searched.

        Returns:
            Tensor: (B, N, 3) L2 distance (NOT squared) to the 3 neighbours.
            Tensor: (B, N, 3) int32 indices of the neighbours.
        """
        assert target.is_contiguous()
        assert source.is_contiguous()

        B, N, _ = target.size()
        m = source.size(1)
        dist2 = torch.empty(B, N, 3, dtype=torch.float32, device=target.device)
        idx = torch.empty(B, N, 3, dtype=torch.int32, device=target.device)
        call("mvp_three_nn", target.device, B, N, m, target, source, dist2, idx)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply
