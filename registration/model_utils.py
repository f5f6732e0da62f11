"""Building blocks of the registration models that touch the op layer --
counterpart of registration/models/dcp.py:35-66 (knn, get_graph_feature) and of
the SVD head shared by dcp.py:331-376 and registration/model_utils.py:213-255.

On float32 CUDA tensors the neighbour search is the knn operator (coordinates)
or Gram matrix + mvp_topk_gram (features), the gather is the grouping operator
and the 3x3 SVDs are ONE mvp_kabsch_svd3 launch instead of a Python loop with a
host synchronisation per sample.  Other tensors (float64 equivalence tests) use
the reference's PyTorch formulation."""
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from mvp_benchmark_amd.mm3d_pn2 import grouping_operation, knn as knn_op  # noqa: E402
from mvp_benchmark_amd.mm3d_pn2.functional import gram_topk  # noqa: E402
from mvp_benchmark_amd.registration import kabsch_rotation  # noqa: E402


def _on_op_layer(t):
    return t.is_cuda and t.dtype == torch.float32


def knn(x, k):
    """x (B,C,N) -> indices (B,N,k) of the k nearest columns (self included),
    nearest first (dcp.py:35-42: topk of -|xi - xj|^2)."""
    if _on_op_layer(x) and x.size(1) == 3 and 0 < k <= 100:
        xt = x.detach().transpose(1, 2).contiguous()
        return knn_op(k, xt, xt, False).transpose(1, 2).contiguous().long()      # (B,k,N) -> (B,N,k)
    if _on_op_layer(x) and 0 < k <= min(64, x.size(2)) and x.size(2) <= 16384:
        xd = x.detach()
        dot = torch.matmul(xd.transpose(2, 1), xd).contiguous()
        return gram_topk(dot, (xd * xd).sum(dim=1).contiguous(), k).long()
    inner = -2 * torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x ** 2, dim=1, keepdim=True)
    pairwise = -xx - inner - xx.transpose(2, 1)
    return pairwise.topk(k=k, dim=-1)[1]


def get_graph_feature(x, k=20, k_major=False):
    """x (B,C,N) -> (B,2C,N,k) = [neighbour, centre] (dcp.py:45-66: the
    neighbour itself, not its offset from the centre).  k_major: (B,2C,k,N) instead -- for the
    1x1 convolution stack of DGCNN the layout is free, and the max over the neighbours then
    reduces a strided dimension with N contiguous instead of 20-element rows (PyTorch's reduce
    kernel runs those at ~0.6 TB/s: 14.7 of 91 ms of the cfg-5 forward, profiles/r3_bench_dcp.txt)."""
    idx = knn(x, k)
    if k_major:
        if _on_op_layer(x):
            nbr = grouping_operation(x.contiguous(), idx.int().transpose(1, 2).contiguous())   # (B,C,k,N)
        else:
            nbr = get_graph_feature(x, k)[:, :x.size(1)].permute(0, 1, 3, 2)
        return torch.cat((nbr, x.unsqueeze(2).expand(-1, -1, k, -1)), dim=1)
    if _on_op_layer(x):
        nbr = grouping_operation(x.contiguous(), idx.int().contiguous())   # (B,C,N,k)
    else:
        b, c, n = x.shape
        base = torch.arange(b, device=x.device).view(-1, 1, 1) * n
        nbr = x.transpose(2, 1).reshape(b * n, c)[(idx + base).view(-1)].view(b, n, k, c).permute(0, 3, 1, 2)
    ctr = x.unsqueeze(3).expand(-1, -1, -1, k)
    return torch.cat((nbr, ctr), dim=1)


def procrustes(src, src_corr, weights=None):
    """Rigid motion taking src (B,3,N) onto src_corr (B,3,N): the SVD head
    (dcp.py:345-376; with `weights` (B,1,N): model_utils.py:221-255).
    Returns R (B,3,3), t (B,3)."""
    src_mean = src.mean(dim=2, keepdim=True)
    corr_mean = src_corr.mean(dim=2, keepdim=True)
    src_centered = src - src_mean
    corr_centered = src_corr - corr_mean
    if weights is not None:
        src_centered = src_centered * weights
    H = torch.matmul(src_centered, corr_centered.transpose(2, 1))
    if _on_op_layer(H):
        R = kabsch_rotation(H)
    else:   # reference formulation, batched
        U, _, Vh = torch.linalg.svd(H)
        V = Vh.transpose(1, 2)
        d = torch.ones(H.shape[0], 3, dtype=H.dtype, device=H.device)
        d[:, 2] = torch.where(torch.linalg.det(V @ U.transpose(1, 2)) < 0, -1.0, 1.0).to(H.dtype)
        R = (V * d.unsqueeze(1)) @ U.transpose(1, 2)
    if weights is None:
        t = torch.matmul(-R, src_mean) + corr_mean
    else:
        t = torch.matmul(-R, (weights * src).sum(dim=2, keepdim=True)) + (weights * src_corr).sum(dim=2, keepdim=True)
    return R, t.view(src.size(0), 3)
