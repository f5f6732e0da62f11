"""Pairwise squared feature distance -- counterpart of the reference's
utils/mm3d_pn2/ops/furthest_point_sample/utils.py:4-31."""
import torch


def calc_square_dist(point_feat_a, point_feat_b, norm=True):
    """
    Args:
        point_feat_a (Tensor): (B, N, C) feature vector of each point.
        point_feat_b (Tensor): (B, M, C) feature vector of each point.
        norm (bool): divide the (rooted) distance by C. Default: True.

    Returns:
        Tensor: (B, N, M) distance between each pair of points.
    """
    num_channel = point_feat_a.shape[-1]
    a_square = point_feat_a.pow(2).sum(dim=-1, keepdim=True)   # (B, N, 1)
    b_square = point_feat_b.pow(2).sum(dim=-1).unsqueeze(1)    # (B, 1, M)
    coor = torch.matmul(point_feat_a, point_feat_b.transpose(1, 2))
    dist = a_square + b_square - 2 * coor
    if norm:
        dist = torch.sqrt(dist) / num_channel
    return dist
