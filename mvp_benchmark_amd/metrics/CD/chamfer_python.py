"""Device-agnostic PyTorch Chamfer distance -- counterpart of the reference's
only CPU path for this op layer, utils/metrics/CD/chamfer_python.py:18-39
(``distChamfer``): cast to float64, P = |x|^2 + |y|^2 - 2 x.y^T, min over both
axes, cast back to float32 / int32.

Same arithmetic in the same order (so results are bit-identical to the
reference on the same torch build), but evaluated in batch chunks so that the
B x N x M float64 matrix never has to exist at once (unchunked B=64,
16384 x 16384 would need 137 GB).  Used as the timed CPU baseline for CD.
"""
import torch


def distChamfer(a, b, chunk=None):
    """
    :param a: (B, N, D) point clouds
    :param b: (B, M, D) point clouds
    :param chunk: clouds per evaluation chunk (None = whole batch at once)
    :return: dist a->b (B,N) f32, dist b->a (B,M) f32, idx (B,N) i32,
             idx (B,M) i32
    """
    bs = a.size(0)
    step = bs if not chunk else int(chunk)
    d1, d2, i1, i2 = [], [], [], []
    for s in range(0, bs, step):
        x = a[s:s + step].double()
        y = b[s:s + step].double()
        xx = torch.pow(x, 2).sum(2)               # (b, N)
        yy = torch.pow(y, 2).sum(2)               # (b, M)
        zz = torch.bmm(x, y.transpose(2, 1))      # (b, N, M)
        P = xx.unsqueeze(2) + yy.unsqueeze(1) - 2 * zz
        m2 = torch.min(P, 2)
        m1 = torch.min(P, 1)
        d1.append(m2[0].float())
        d2.append(m1[0].float())
        i1.append(m2[1].int())
        i2.append(m1[1].int())
    return torch.cat(d1), torch.cat(d2), torch.cat(i1), torch.cat(i2)


def pairwise_dist(x, y):
    """(N,D),(M,D) -> (N,M) squared distances via the Gram-matrix identity
    (chamfer_python.py:4-9)."""
    xx = (x * x).sum(1)
    yy = (y * y).sum(1)
    return xx.unsqueeze(1) + yy.unsqueeze(0) - 2 * torch.mm(x, y.t())


def NN_loss(x, y, dim=0):
    """Mean nearest-neighbour squared distance (chamfer_python.py:12-15)."""
    values, _ = pairwise_dist(x, y).min(dim=dim)
    return values.mean()
