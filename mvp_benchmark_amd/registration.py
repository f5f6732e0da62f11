"""Batched Kabsch / Procrustes rotation for the DCP registration head.

`kabsch_rotation(H)` replaces the per-sample loop of the reference's SVDHead
(registration/models/dcp.py:360-373, registration/model_utils.py:229-240):

    for i in range(B):
        u, s, v = torch.svd(H[i]); r = v @ u.T
        if det(r) < 0: v = v @ diag(1, 1, -1); r = v @ u.T

B tiny LAPACK-style calls with a host synchronisation each become ONE launch of
mvp_kabsch_svd3 (one lane per matrix, float64 one-sided Jacobi).  Differentiable:
the backward pass is the closed-form SVD adjoint on (B,3,3) tensors (batched
elementwise / 3x3 matmul PyTorch ops, device-agnostic, no loop, no sync) -- the
same gradient torch.svd's autograd produces for R = V D U^T with the reflection
D held fixed.
"""
import torch
from torch.autograd import Function

from ._lib import call


def svd3_kabsch_backward(U, S, V, flipped, grad_R):
    """Adjoint of H -> R = V diag(1,1,d) U^T  (H = U diag(S) V^T, d = -1 where
    `flipped`) for batches of 3x3 matrices: returns grad_H (B,3,3).

    dR = dV D U^T + V D dU^T  =>  gU = gR^T V D,  gV = gR U D,  gS = 0, then the
    SVD adjoint for square full-rank A (Townsend 2016; the formula behind
    torch.linalg.svd's autograd):
        gA = U [ (skew(U^T gU) / E) S + S (skew(V^T gV) / E) ] V^T,
        skew(X) = X - X^T,  E_ij = s_j^2 - s_i^2 (i != j), E_ii = 1.
    Singular where two singular values coincide, exactly like torch.svd."""
    d = torch.ones_like(S)
    d[:, 2] = torch.where(flipped.bool(), -torch.ones_like(S[:, 2]), torch.ones_like(S[:, 2]))
    VD = V * d.unsqueeze(1)                       # V diag(d)
    UD = U * d.unsqueeze(1)
    gU = grad_R.transpose(1, 2) @ VD
    gV = grad_R @ UD
    s2 = S * S
    E = s2.unsqueeze(1) - s2.unsqueeze(2)         # E_ij = s_j^2 - s_i^2
    eye = torch.eye(3, dtype=S.dtype, device=S.device)
    E = E + eye
    Su = U.transpose(1, 2) @ gU
    Sv = V.transpose(1, 2) @ gV
    Ju = (Su - Su.transpose(1, 2)) / E * (1 - eye)
    Jv = (Sv - Sv.transpose(1, 2)) / E * (1 - eye)
    inner = Ju * S.unsqueeze(1) + S.unsqueeze(2) * Jv
    return U @ inner @ V.transpose(1, 2)


class KabschSVD(Function):
    """H (B,3,3) float32 CUDA -> R (B,3,3): mvp_kabsch_svd3."""

    @staticmethod
    def forward(ctx, H):
        assert H.dim() == 3 and H.shape[1:] == (3, 3)
        H = H.contiguous().float()
        b = H.shape[0]
        R = torch.empty_like(H)
        U = torch.empty_like(H)
        V = torch.empty_like(H)
        S = torch.empty(b, 3, device=H.device, dtype=torch.float32)
        flipped = torch.empty(b, device=H.device, dtype=torch.int32)
        call("mvp_kabsch_svd3", H.device, b, H, R, U, S, V, flipped)
        ctx.save_for_backward(U, S, V, flipped)
        return R

    @staticmethod
    def backward(ctx, grad_R):
        U, S, V, flipped = ctx.saved_tensors
        return svd3_kabsch_backward(U, S, V, flipped, grad_R.contiguous())


kabsch_rotation = KabschSVD.apply


def svd3(H):
    """(U, S, V, R, flipped) of a batch of 3x3 matrices (no autograd): the
    factors torch.svd would return plus the Kabsch rotation."""
    H = H.contiguous().float()
    b = H.shape[0]
    R, U, V = torch.empty_like(H), torch.empty_like(H), torch.empty_like(H)
    S = torch.empty(b, 3, device=H.device, dtype=torch.float32)
    flipped = torch.empty(b, device=H.device, dtype=torch.int32)
    call("mvp_kabsch_svd3", H.device, b, H, R, U, S, V, flipped)
    return U, S, V, R, flipped
