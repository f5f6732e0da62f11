"""k nearest neighbours -- same surface as the reference's
utils/mm3d_pn2/ops/knn/knn.py:7-72, backed by mvp_knn."""
import torch
from torch.autograd import Function

from ...._lib import call


class KNN(Function):
    """Heap-based kNN (ascending by squared distance, k <= 100)."""

    @staticmethod
    def forward(ctx,
                k: int,
                xyz: torch.Tensor,
                center_xyz: torch.Tensor = None,
                transposed: bool = False) -> torch.Tensor:
        """
        Args:
            k (int): number of nearest neighbours.
            xyz (Tensor): (B, N, 3), or (B, 3, N) if transposed.
            center_xyz (Tensor): (B, npoint, 3), or (B, 3, npoint) if
                transposed; defaults to xyz.
            transposed (bool): inputs are channel-first. Pass positionally
                (knn = KNN.apply takes no keywords).

        Returns:
            Tensor: (B, k, npoint) int32 indices.
        """
        assert k > 0

        if center_xyz is None:
            center_xyz = xyz

        if transposed:
            xyz = xyz.transpose(2, 1).contiguous()
            center_xyz = center_xyz.transpose(2, 1).contiguous()

        assert xyz.is_contiguous()  # [B, N, 3]
        assert center_xyz.is_contiguous()  # [B, npoint, 3]
        assert center_xyz.device == xyz.device, \
            'center_xyz and xyz should be put on the same device'

        B, npoint, _ = center_xyz.shape
        N = xyz.shape[1]

        idx = center_xyz.new_zeros((B, npoint, k)).int()
        dist2 = center_xyz.new_zeros((B, npoint, k)).float()
        call("mvp_knn", xyz.device, B, N, npoint, k, xyz, center_xyz, idx,
             dist2)
        # idx shape to [B, k, npoint]
        idx = idx.transpose(2, 1).contiguous()
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


knn = KNN.apply
