"""NaiveSyncBatchNorm1d/2d -- name-compatible counterparts of the reference's
utils/mm3d_pn2/ops/norm.py:27-133 (the only explicit collective call site of
the reference, :17,23).  No completion model uses them (SURVEY.md section 2,
row 17); they are provided so ``from mm3d_pn2 import NaiveSyncBatchNorm1d``
keeps working.  Statistics are synchronised with one all_reduce of the
per-channel (sum, sum of squares, count) over RCCL; autograd flows through
torch.distributed.nn-free math (the reduced statistics are treated as in the
reference: mean/var are shared by every rank)."""
import torch
import torch.distributed as dist
from torch import nn as nn
from torch.autograd.function import Function


class _AllReduceSum(Function):
    """all_reduce(SUM) whose backward is all_reduce(SUM) (norm.py:9-24 uses
    all_gather + sum forward / all_reduce backward, which is the same map)."""

    @staticmethod
    def forward(ctx, input):
        out = input.clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        g = grad_output.clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return g


def _world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _sync_bn(bn, input, reduce_dims, view):
    if _world_size() == 1 or not bn.training:
        return None
    assert input.shape[0] > 0, 'SyncBN does not support empty inputs'
    C = input.shape[1]
    mean = torch.mean(input, dim=reduce_dims)
    meansqr = torch.mean(input * input, dim=reduce_dims)
    vec = torch.cat([mean, meansqr], dim=0)
    vec = _AllReduceSum.apply(vec) * (1.0 / _world_size())
    mean, meansqr = torch.split(vec, C)
    var = meansqr - mean * mean
    bn.running_mean += bn.momentum * (mean.detach() - bn.running_mean)
    bn.running_var += bn.momentum * (var.detach() - bn.running_var)
    invstd = torch.rsqrt(var + bn.eps)
    scale = bn.weight * invstd
    bias = bn.bias - mean * scale
    return input * scale.reshape(view) + bias.reshape(view)


class NaiveSyncBatchNorm1d(nn.BatchNorm1d):
    """BatchNorm1d whose batch statistics are averaged over all ranks
    (equal per-rank batch sizes assumed, as in the reference)."""

    def forward(self, input):
        if input.dim() == 3:
            out = _sync_bn(self, input, [0, 2], (1, -1, 1))
        else:
            out = _sync_bn(self, input, [0], (1, -1))
        return super().forward(input) if out is None else out


class NaiveSyncBatchNorm2d(nn.BatchNorm2d):
    """BatchNorm2d whose batch statistics are averaged over all ranks."""

    def forward(self, input):
        out = _sync_bn(self, input, [0, 2, 3], (1, -1, 1, 1))
        return super().forward(input) if out is None else out
