"""Per-point / per-edge linear maps (1x1 convolutions) of the completion
networks.  Forward and data gradient are the library convolution (MIOpen runs
them at 1.3-2.7 TB/s of their operands); the weight gradient of layers with few
output channels -- a GEMM with a tiny output and everything else as its
reduction dimension, which MIOpen runs at 0.4-1.0 TB/s -- goes through
`mvp_pointwise_wgrad` (include/mvpops.h).  Same parameters and state_dict layout
as nn.Conv1d / nn.Conv2d(kernel_size=1); the reference builds these layers with
nn.Conv1d / nn.Conv2d directly (completion/models/pcn.py, ecg.py, vrcnet.py).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from ._lib import call, pointwise_wgrad_scratch_bytes

MAX_COUT = 64   # the kernel's limit
MAX_CIN = 64    # beyond this MIOpen's weight gradient is as fast or faster (tools/bench_conv_parts.py)


def _covered(x, weight):
    if not (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() in (3, 4)):
        return False
    if x.numel() == 0 or weight.numel() == 0:
        return False
    length = x[0, 0].numel()
    return weight.size(0) <= MAX_COUT and weight.size(1) <= MAX_CIN and length % 4 == 0 and length > 0 \
        and x.size(0) <= 65535


class _PointwiseConv(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        conv = F.conv1d if x.dim() == 3 else F.conv2d
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return conv(x, weight, bias)

    @staticmethod
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        cout, cin = weight.shape[:2]
        nd = x.dim() - 2
        gy = grad_out.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.ops.aten.convolution_backward(gy, x, weight, None, [1] * nd, [0] * nd, [1] * nd, False, [0] * nd, 1,
                                                     [True, False, False])[0]
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            B = x.size(0)
            length = x[0, 0].numel()
            nbytes = pointwise_wgrad_scratch_bytes(B, cin, cout, length)
            scratch = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            gw = torch.empty_like(weight)
            gb = torch.empty(cout, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            call("mvp_pointwise_wgrad", x.device, B, cin, cout, length, x, gy, gw, gb, scratch, nbytes)
        return gx, gw, gb


def pointwise_conv(x, weight, bias=None):
    """y = W x + bias over the channel dimension of x (B,Cin,N) / (B,Cin,H,W);
    weight (Cout,Cin,1[,1])."""
    conv = F.conv1d if x.dim() == 3 else F.conv2d
    if torch.is_grad_enabled() and weight.requires_grad and _covered(x, weight) and x.is_contiguous() \
            and weight.is_contiguous():
        return _PointwiseConv.apply(x, weight, bias)
    return conv(x, weight, bias)


class PointwiseConv1d(nn.Conv1d):
    def __init__(self, c_in, c_out, bias=True):
        super().__init__(c_in, c_out, kernel_size=1, bias=bias)

    def forward(self, x):
        return pointwise_conv(x, self.weight, self.bias)


class PointwiseConv2d(nn.Conv2d):
    def __init__(self, c_in, c_out, bias=True):
        super().__init__(c_in, c_out, kernel_size=1, bias=bias)

    def forward(self, x):
        return pointwise_conv(x, self.weight, self.bias)
