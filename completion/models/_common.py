"""Pieces shared by the three completion networks: layer factories and the
loss / metric tail of Model.forward (identical in the reference's pcn.py
:93-112, ecg.py:233-253 and vrcnet.py:519-526, restated once here)."""
import torch
import torch.nn as nn

from model_utils import calc_cd, calc_emd
from mvp_benchmark_amd.pointwise import PointwiseConv1d, PointwiseConv2d, pointwise_conv


def pointwise1d(c_in, c_out, bias=True):
    """Per-point linear map on (B, C, N) features: an nn.Conv1d(kernel_size=1)
    (same parameters / state_dict) whose weight gradient, for few channels, comes
    from the op layer."""
    return PointwiseConv1d(c_in, c_out, bias=bias)


def pointwise2d(c_in, c_out, bias=True):
    """Per-edge linear map on (B, C, N, k) / (B, C, 1, N) features (nn.Conv2d(kernel_size=1))."""
    return PointwiseConv2d(c_in, c_out, bias=bias)


dense = nn.Linear


def conv_global_concat(conv, global_vec, feats, relu=False, global_first=True):
    """conv(cat((global_vec tiled over the positions, feats), 1)) [then ReLU] without the concatenation: a 1x1
    convolution is linear, so the global feature's share of the product is ONE vector per cloud,
        W [g; f](b, :, n) = W[:, :Cg] g(b) + bias  +  W[:, Cg:] f(b, :, n),
    computed once (a (B, Cg) x (Cg, Cout) product) and added to the convolution of the per-point channels --
    the reference tiles the vector N times and pays Cg * Cout multiply-adds per point for it (vrcnet.py conv6,
    ecg.py conv5, pcn.py conv3).  conv: a PointwiseConv1d / 2d over Cg + Cf channels, global_vec (B, Cg), feats
    (B, Cf, N) / (B, Cf, 1, N).  Same parameters, same function up to float32 summation order."""
    cout, cg = conv.out_channels, global_vec.size(1)
    w = conv.weight.view(cout, -1)
    wg, wf = (w[:, :cg], w[:, cg:]) if global_first else (w[:, w.size(1) - cg:], w[:, :w.size(1) - cg])
    per_cloud = nn.functional.linear(global_vec, wg, conv.bias)
    tail = (1,) * (feats.dim() - 2)
    h = pointwise_conv(feats, wf.contiguous().view(cout, -1, *tail)) + per_cloud.view(per_cloud.shape + tail)
    return torch.relu_(h) if relu else h


def shape_loss(kind, pred, gt):
    """Per-cloud training loss (B,): 'cd' -> cd_p of calc_cd, 'emd' -> calc_emd
    at the training setting (eps 0.005, 50 iterations)."""
    if kind == 'cd':
        return calc_cd(pred, gt)[0]
    if kind == 'emd':
        return calc_emd(pred, gt)
    raise NotImplementedError('Train loss is either CD or EMD!')


def eval_outputs(coarse, fine, gt, eval_emd):
    """prefix == "val": the metric dictionary train.py / test.py consume.  EMD
    runs at the evaluation setting (eps 0.004, 3000 iterations) when enabled."""
    emd = calc_emd(fine, gt, eps=0.004, iterations=3000) if eval_emd else 0
    cd_p, cd_t, f1 = calc_cd(fine, gt, calc_f1=True)
    return {'out1': coarse, 'out2': fine, 'emd': emd, 'cd_p': cd_p, 'cd_t': cd_t, 'f1': f1}
