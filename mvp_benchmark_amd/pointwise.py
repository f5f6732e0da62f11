"""Per-point / per-edge linear maps (1x1 convolutions) of the completion
networks.  Same parameters and state_dict layout as nn.Conv1d /
nn.Conv2d(kernel_size=1); the reference builds these layers with nn.Conv1d /
nn.Conv2d directly (completion/models/pcn.py, ecg.py, vrcnet.py).

Routing on float32 CUDA tensors (include/mvpops.h), training and inference alike:
  * layers with >= 32 input and output channels: all three passes on the float32
    MFMA GEMMs of csrc/pointwise_mfma.hip -- forward `mvp_pointwise_mfma` with bias /
    ReLU / residual / max-over-neighbours fused into its epilogue, data gradient
    the same kernel on the transposed weight with ReLU' applied to grad_out on
    load, weight + bias gradient `mvp_pointwise_wgrad_mfma` (ReLU' on load, the
    NCHW operands read as they lie: no layout transposes);
  * layers with <= 64 x 64 channels: weight gradient `mvp_pointwise_wgrad`, data
    gradient `mvp_pointwise_dgrad` (one pass over gy, weights from LDS);
  * everything else (and every non-CUDA / non-float32 tensor): the library
    convolution.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from ._lib import call, pointwise_max_backward_scratch_bytes, pointwise_wgrad_mfma_scratch_bytes, pointwise_wgrad_scratch_bytes

MAX_COUT = 64   # mvp_pointwise_wgrad's limit
MAX_CIN = 64
# Channel minimum of the MFMA route.  Rounds 2-3: 32 (below it a 32-wide MFMA block is mostly padding).  Round 4: 1 -- the
# skinny layers (attention-weight MLPs 544 / 272 / 136 / 68 -> 16 / 8 / 4 / 2, the 3- and 8-channel inputs) move bytes, not
# flops, and the library needs 25-200 us per pass for them where one read of the activations takes 7-25
# (profiles/r4_conv_passes_vrcnet_skinny.txt: 3.60 -> 1.97 ms per step over the 16 shapes).
MFMA_MIN_CH = 1
MFMA_SKINNY_FWD_MAX_CIN = 136   # forward with < 32 output channels: the library's GEMV-like kernel wins from ~256 input channels
# Round 4 (tools/bench_conv_passes.py, profiles/r4_conv_passes_*.txt): on the 20 routed shapes of a VRCNet step the
# three passes cost 3.99 / 3.93 / 5.38 ms on these kernels against 5.34 / 5.54 / 6.29 ms on the library (whose
# weight gradient is an NHWC implicit GEMM wrapped in layout transposes), every shape at least a tie: every routed layer
# trains and infers on the MFMA kernels.  What is left below are ROUTE SELECTORS, not experiments: the parity tests drive
# every branch of the backward pass through them (tests/test_gpu_harness.py) and the reference-formulation report and the
# per-pass bench record the library route beside ours (tests/report_reference_model_step.py, tools/bench_conv_passes.py).
# (Round 6 removed the superseded ones: MFMA_WGRAD_TRAIN / _MIN_CIN -- weight gradients only --, MFMA_FWD_MAX_CIN.)
USE_MFMA = True            # False: nothing is routed to the MFMA kernels (the library's convolution everywhere)
MFMA_TRAIN = True          # under autograd the routed layers go through _PointwiseConv
MFMA_DGRAD = True          # data gradient on mvp_pointwise_mfma (W^T, ReLU' on load)
MFMA_WGRAD_MIN_CIN = 1     # weight gradient on mvp_pointwise_wgrad_mfma from this many input channels
# Round 6: what "the library" is for a float32 CUDA layer our kernels leave to it -- batched GEMMs (rocBLAS / hipBLASLt
# through torch.matmul), not MIOpen's convolutions: a 1x1 convolution IS a GEMM per cloud; MIOpen wraps it in NHWC
# transposes, starts every process on its naive reference kernels (5.4 ms per weight gradient of ECG's bottleneck layers for
# the first ~150 steps) until its solver search settles on a different kernel from run to run, and its implicit-GEMM
# backward-data kernel reads out of bounds on odd shapes (profiles/r6_miopen_igemm_bwd_fault.txt; that one is avoided either
# way: _PointwiseConv.backward).  Measured (tools/ab_flag.py, 12 alternations, ECG -- the model with library-routed layers):
# first alternation 23.9 -> 18.5 ms (no naive-kernel phase), steady state 16.94 = 16.93 ms in rounds 3-6 but 17.2 -> 17.55 in
# rounds 8-12; VRCNet 18.39 -> 18.45.  Not faster, so the default stays MIOpen for the forward and the weight gradient of
# those layers; True = no MIOpen call at all in the training path.
LIBRARY_IS_GEMM = False

_WGRAD_SCRATCH = {}   # (device, stream) -> one growing workspace for the partial tiles (no allocator churn)


def _wgrad_scratch(device, nbytes):
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _WGRAD_SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _WGRAD_SCRATCH[key] = buf
    return buf


MFMA_WGRAD_MIN_POSITIONS = 16384   # B * L below this: too few slabs of positions to deal out (ECG's 64- and 256-point levels)


def _gemm_fits(batch, m, k, length):
    """Per-cloud GEMM (m x k) (k x length) worth running on mvp_pointwise_mfma?  Not with a long
    reduction and fewer workgroups (128 x 128 tiles of the output) than the chip holds at once, or a
    column tile that is half empty: the library splits K there (ECG's bottleneck layers: 1864 -> 768 at
    256 points 0.28 against 0.21 ms, 2824 -> 1024 at 64 points 0.25 against 0.10;
    profiles/r4_conv_passes_ecg.txt)."""
    if k <= 512:   # (round 6: 512 included -- (64, 512 -> 512, 384) 0.127 against the library's 0.145 ms, the data gradient of
        return True   # (64, 128 -> 512, 384) 0.050 against 0.108: tools/bench_conv_passes.py)
    if length < 128:
        return False
    bm = 128 if m > 64 else 64
    return -(-m // bm) * -(-length // 128) * batch >= 512


def _mfma_fwd(x, cin, cout, weight=None):
    """Forward GEMM through mvp_pointwise_mfma?"""
    return _mfma_ok(x, cin, cout, weight) \
        and _gemm_fits(x.size(0), cout, cin, x[0, 0].numel()) and (cout >= 32 or cin <= MFMA_SKINNY_FWD_MAX_CIN)


def _mfma_ok(x, cin, cout, weight=None):
    """Shape / layout covered by the MFMA kernels?  (`weight`: the kernels read it with 16-byte
    loads -- a parameter that is a view at an odd offset of a flattened / bucketed storage falls
    back to the library convolution instead of failing with MVP_EBADARG.)"""
    if not (USE_MFMA and x.is_cuda and x.dtype == torch.float32 and x.dim() in (3, 4) and x.is_contiguous()):
        return False
    if weight is not None and (weight.data_ptr() % 16 != 0 or not weight.is_contiguous()):
        return False
    length = x[0, 0].numel() if x.numel() else 0
    return cin >= MFMA_MIN_CH and cout >= MFMA_MIN_CH and length > 0 and length % 4 == 0 and x.size(0) <= 65535 \
        and x.data_ptr() % 16 == 0


PW_RELU, PW_RELU_AFTER, PW_RES_IS_MASK, PW_X_RELU = 1, 2, 4, 8     # include/mvpops.h MVP_PW_*


def mfma_linear(x, w2d, bias=None, relu=False, residual=None, group=1, w_kmajor=False, xmask=None, x_relu=False,
                relu_after=False, res_is_mask=False, bias_per_cloud=False, m_split=0):
    """y = epilogue(W x) for x (B, Cin, ...) contiguous float32 CUDA, W = w2d (Cout, Cin) -- or its
    transpose (Cin, Cout) with w_kmajor; with xmask, x counts as 0 where xmask <= 0; with x_relu where it is
    <= 0 itself.  Epilogue (mvp_pointwise_mfma_ex): + bias (one per cloud with bias_per_cloud: (B, Cout)), ReLU,
    max over groups of `group` consecutive positions of the flattened trailing dimensions, + residual (or, with
    res_is_mask, zero where residual <= 0), ReLU again with relu_after.  m_split > 0: two outputs, the rows below
    m_split and the others -> (y, y2).  No autograd."""
    B = x.size(0)
    cin = x.size(1)
    cout = w2d.size(1) if w_kmajor else w2d.size(0)
    length = x[0, 0].numel()
    y2 = None
    if m_split:
        y = torch.empty((B, m_split) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        y2 = torch.empty((B, cout - m_split) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    elif group == 1:
        y = torch.empty((B, cout) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    else:
        y = torch.empty(B, cout, length // group, dtype=torch.float32, device=x.device)
    ldw = 0
    if not w_kmajor and cin % 4 != 0:
        # rows padded to a multiple of 4 floats: the kernel reads the weight with 16-byte loads (PCN's 1029 -> 512
        # folding layer at 16384 points: 5.49 -> 4.7 ms; the copy is 2 MB)
        w2d = F.pad(w2d, (0, -cin % 4))
        ldw = w2d.size(1)
    flags = (PW_RELU if relu else 0) | (PW_RELU_AFTER if relu_after else 0) | (PW_RES_IS_MASK if res_is_mask else 0) \
        | (PW_X_RELU if x_relu else 0)
    call("mvp_pointwise_mfma_ex", x.device, B, cin, cout, length, x, xmask, w2d, ldw, int(w_kmajor), bias, int(bias_per_cloud),
         residual, flags, int(group), y, int(m_split), y2)
    return (y, y2) if m_split else y


def mfma_wgrad(x, gy, cout, cin, with_bias, gymask=None, x_relu=False):
    """(gw (Cout, Cin), gb (Cout) | None) of y = W x + b from x (B, Cin, ...) and gy (B, Cout, ...);
    with gymask, gy counts as 0 where gymask <= 0; with x_relu, x as 0 where it is <= 0 (mvp_pointwise_wgrad_mfma_ex)."""
    B = x.size(0)
    length = x[0, 0].numel()
    nbytes = pointwise_wgrad_mfma_scratch_bytes(B, cin, cout, length, with_bias)
    scratch = _wgrad_scratch(x.device, nbytes)     # stream-ordered reuse: the reduce kernel has read it before the next call writes
    gw = torch.empty(cout, cin, dtype=torch.float32, device=x.device)
    gb = torch.empty(cout, dtype=torch.float32, device=x.device) if with_bias else None
    call("mvp_pointwise_wgrad_mfma_ex", x.device, B, cin, cout, length, x, int(x_relu), gy, gymask, gw, gb, scratch, nbytes)
    return gw, gb


def _covered(x, weight):
    if not (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() in (3, 4)):
        return False
    if x.numel() == 0 or weight.numel() == 0:
        return False
    length = x[0, 0].numel()
    return weight.size(0) <= MAX_COUT and weight.size(1) <= MAX_CIN and length % 4 == 0 and length > 0 \
        and x.size(0) <= 65535


def _library_conv(x, weight, bias):
    """The layer on the library, no autograd: conv1d / conv2d (MIOpen) or, with LIBRARY_IS_GEMM on float32 CUDA tensors, one
    batched GEMM W x[b] (+ bias through baddbmm's addend)."""
    if LIBRARY_IS_GEMM and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32:
        cout, cin = weight.shape[:2]
        x3 = x.flatten(2)
        w3 = weight.reshape(1, cout, cin).expand(x.size(0), cout, cin)
        y = torch.bmm(w3, x3) if bias is None else torch.baddbmm(bias.view(1, cout, 1), w3, x3)
        return y.view((x.size(0), cout) + tuple(x.shape[2:]))
    return (F.conv1d if x.dim() == 3 else F.conv2d)(x, weight, bias)


class _PointwiseConv(Function):
    """y = [relu](W x + bias): forward and data gradient on the MFMA GEMM where the
    shape allows, the small-channel weight gradient through mvp_pointwise_wgrad, the
    rest through the library's convolution passes."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        conv = F.conv1d if x.dim() == 3 else F.conv2d
        cout, cin = weight.shape[:2]
        ctx.has_bias = bias is not None
        ctx.relu = relu
        if not x.is_contiguous():          # (a permuted view: the kernels of the backward pass take dense tensors)
            x = x.contiguous()
        if not weight.is_contiguous():
            weight = weight.contiguous()
        if (MFMA_TRAIN or _covered(x, weight)) and _mfma_fwd(x, cin, cout, weight):
            y = mfma_linear(x, weight.view(cout, cin), bias, relu=relu)
        else:
            y = _library_conv(x, weight, bias)
            if relu:
                y = torch.relu_(y)
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, grad_out):
        x, weight, y = ctx.saved_tensors
        cout, cin = weight.shape[:2]
        nd = x.dim() - 2
        gy = grad_out.contiguous()
        need_x = ctx.needs_input_grad[0]
        need_w = ctx.needs_input_grad[1]
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        gx_small = need_x and _covered(x, weight)          # <= 64 x 64 channels: mvp_pointwise_dgrad (11-26 us, at or below the MFMA kernel)
        gx_mfma = MFMA_DGRAD and need_x and not gx_small and _mfma_ok(gy, cout, cin, weight) and cin % 4 == 0 \
            and _gemm_fits(x.size(0), cin, cout, x[0, 0].numel())
        gw_mfma = (need_w or need_b) and cin >= MFMA_WGRAD_MIN_CIN and _mfma_ok(x, cin, cout, weight) and not _covered(x, weight) \
            and x.size(0) * x[0, 0].numel() >= MFMA_WGRAD_MIN_POSITIONS \
            and pointwise_wgrad_mfma_scratch_bytes(x.size(0), cin, cout, x[0, 0].numel(), ctx.has_bias) > 0
        # ReLU'(.): the MFMA kernels mask grad_out by the saved output on load; the other routes get
        # the masked tensor
        mask = y if ctx.relu else None
        if ctx.relu and ((need_x and not gx_mfma) or ((need_w or need_b) and not gw_mfma)):
            gy_masked = gy * (y > 0)
        else:
            gy_masked = gy
        gx = gw = gb = None
        if need_x:
            if gx_mfma:
                gx = mfma_linear(gy, weight.view(cout, cin), w_kmajor=True, xmask=mask)       # W^T (gy . relu')
            elif _covered(x, weight) and gy_masked.data_ptr() % 16 == 0:
                # few channels: W^T gy in one pass over gy (mvp_pointwise_dgrad) -- not the library's implicit-GEMM
                # kernel, one variant of which reads out of bounds on these shapes (csrc/pointwise.hip)
                gx = torch.empty_like(x)
                call("mvp_pointwise_dgrad", x.device, x.size(0), cin, cout, x[0, 0].numel(), weight, gy_masked, gx)
            else:
                # W^T gy per cloud as a batched library GEMM -- NOT aten.convolution_backward: MIOpen's implicit-GEMM
                # backward-data kernel (igemm_bwd_gtcx35_nhwc_fp32) reads past its operands on some of these shapes (37- and
                # 50-point layers of the golden tests: a GPU memory fault whenever the neighbouring page happens to be
                # unmapped -- rocgdb trace in profiles/NOTES_r6.md section 11); a 1x1 convolution's data gradient is a GEMM
                gx = torch.matmul(weight.reshape(cout, cin).t(), gy_masked.flatten(2)).view_as(x)
        gy = gy_masked
        if need_w or need_b:
            if gw_mfma:
                gw, gb = mfma_wgrad(x, grad_out.contiguous(), cout, cin, ctx.has_bias, gymask=mask)
                gw = gw.view_as(weight)
            elif _covered(x, weight):
                B = x.size(0)
                length = x[0, 0].numel()
                nbytes = pointwise_wgrad_scratch_bytes(B, cin, cout, length)
                scratch = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
                gw = torch.empty_like(weight)
                gb = torch.empty(cout, dtype=torch.float32, device=x.device) if ctx.has_bias else None
                call("mvp_pointwise_wgrad", x.device, B, cin, cout, length, x, gy, gw, gb, scratch, nbytes)
            elif LIBRARY_IS_GEMM or x[0, 0].numel() % 4 != 0 or not x.is_contiguous():
                # (the library = GEMMs, see LIBRARY_IS_GEMM; MIOpen's weight-gradient kernels only with the switch off)
                gw = torch.einsum("bol,bil->oi", gy.flatten(2), x.flatten(2)).view_as(weight) if need_w else None
                gb = gy.flatten(2).sum((0, 2)) if need_b else None
            else:
                _, gw, gb = torch.ops.aten.convolution_backward(
                    gy, x, weight, [cout] if ctx.has_bias else None, [1] * nd, [0] * nd, [1] * nd, False, [0] * nd, 1,
                    [False, need_w, need_b])
        return gx, gw, gb, None


def pointwise_conv(x, weight, bias=None, relu=False):
    """y = W x + bias (then ReLU if `relu`) over the channel dimension of x
    (B,Cin,N) / (B,Cin,H,W); weight (Cout,Cin,1[,1])."""
    conv = F.conv1d if x.dim() == 3 else F.conv2d
    cout, cin = weight.shape[:2]
    routed = x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.is_contiguous() \
        and weight.is_contiguous() and x.numel() > 0 and (_mfma_ok(x, cin, cout, weight) or _covered(x, weight))
    wants_grad = torch.is_grad_enabled() and (weight.requires_grad or x.requires_grad or (bias is not None and bias.requires_grad))
    if routed and wants_grad:
        if MFMA_TRAIN or _covered(x, weight):
            return _PointwiseConv.apply(x, weight, bias, relu)
        y = conv(x, weight, bias)
        return torch.relu(y) if relu else y
    if wants_grad and USE_MFMA and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.numel() > 0:
        # shapes no kernel of ours covers (positions not a multiple of 4, strided tensors): the library's forward, but STILL
        # _PointwiseConv's backward -- its data gradient is a batched GEMM, never MIOpen's implicit-GEMM backward-data kernel,
        # which reads out of bounds on such shapes (see _PointwiseConv.backward)
        return _PointwiseConv.apply(x, weight, bias, relu)
    if routed and _mfma_fwd(x, cin, cout, weight):                 # inference
        return mfma_linear(x, weight.view(cout, cin), bias, relu=relu)
    y = _library_conv(x, weight, bias) if (USE_MFMA and not wants_grad) else conv(x, weight, bias)
    return torch.relu(y) if relu else y


def _fused_routes(x, weight, need_x):
    """All three passes of a layer on the MFMA kernels (the fused prologues / epilogues live there only)?"""
    cout, cin = weight.shape[:2]
    if not (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.is_contiguous()
            and weight.is_contiguous() and x.numel() > 0 and _mfma_ok(x, cin, cout, weight)):
        return False
    B, length = x.size(0), x[0, 0].numel()
    # (the skinny-forward rule of _mfma_fwd is not applied: 544 -> 16 at 384 points costs 0.031 ms here against the
    # library's 0.024, the passes over the 53 MB input that the fused prologue saves cost 0.06)
    if not _gemm_fits(B, cout, cin, length):
        return False
    if need_x and not (cin % 4 == 0 and _gemm_fits(B, cin, cout, length)):
        return False
    return B * length >= MFMA_WGRAD_MIN_POSITIONS and pointwise_wgrad_mfma_scratch_bytes(B, cin, cout, length, True) > 0


class _PointwiseConvFused(Function):
    """y = act2(act1(W in(x) + bias + cloud_bias[b]) + residual) with in = ReLU if relu_in, act1 = ReLU if relu, act2 = ReLU
    if relu_after, in ONE GEMM (mvp_pointwise_mfma_ex): the pre-activation ReLUs, residual sums and per-cloud vectors of the
    relational encoder (vrcnet.py:34-57, 151, 172, 283-296) cost no elementwise pass forward.  Backward: ONE pass masks
    grad_out by the output where a residual or a per-cloud bias needs the masked tensor itself (otherwise the GEMMs mask on
    load), the data gradient masks its output by x > 0 in its epilogue, the weight gradient takes relu(x) on load."""

    @staticmethod
    def forward(ctx, x, weight, bias, cloud_bias, residual, relu_in, relu, relu_after):
        cout, cin = weight.shape[:2]
        b = bias
        per_cloud = False
        if cloud_bias is not None:
            b = cloud_bias if bias is None else cloud_bias + bias          # (B, Cout): tiny
            per_cloud = True
        y = mfma_linear(x, weight.view(cout, cin), b, relu=relu, residual=residual, x_relu=relu_in, relu_after=relu_after,
                        bias_per_cloud=per_cloud)
        ctx.has_bias, ctx.has_cloud_bias, ctx.has_residual = bias is not None, cloud_bias is not None, residual is not None
        ctx.relu_in, ctx.relu_out = relu_in, relu or relu_after
        ctx.save_for_backward(x, weight, y if ctx.relu_out else None)
        return y

    @staticmethod
    def backward(ctx, grad_out):
        x, weight, y = ctx.saved_tensors
        cout, cin = weight.shape[:2]
        gy = grad_out.contiguous()
        mask = None
        if ctx.relu_out:
            if ctx.has_residual or ctx.has_cloud_bias:
                gy = torch.ops.aten.threshold_backward(gy, y, 0)          # the masked tensor itself is an output
            else:
                mask = y
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        gx = gw = gb = gcb = None
        if need_x:
            gx = mfma_linear(gy, weight.view(cout, cin), w_kmajor=True, xmask=mask,
                             residual=x if ctx.relu_in else None, res_is_mask=ctx.relu_in)
        if need_w or need_b:
            gw, gb = mfma_wgrad(x, gy, cout, cin, ctx.has_bias, gymask=mask, x_relu=ctx.relu_in)
            gw = gw.view_as(weight)
        if ctx.has_cloud_bias and ctx.needs_input_grad[3]:
            gcb = gy.flatten(2).sum(2)
        gres = gy if ctx.has_residual and ctx.needs_input_grad[4] else None
        return gx, gw, gb, gcb, gres, None, None, None


def pointwise_conv_fused(x, weight, bias=None, relu_in=False, relu=False, residual=None, relu_after=False, cloud_bias=None):
    """act2(act1(W in(x) + bias + cloud_bias[b]) + residual): in = ReLU if relu_in, act1 = ReLU if relu, act2 = ReLU if
    relu_after; cloud_bias (B, Cout) (summed with the bias first: one addend per output), residual like the output.  One GEMM where all passes of the layer are on the MFMA
    kernels, the same function composed of pointwise_conv and elementwise passes elsewhere (host tensors, other dtypes,
    shapes the routing rules give to the library)."""
    wants_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, weight, bias, residual, cloud_bias))
    if _fused_routes(x, weight, x.requires_grad and wants_grad) and (residual is None or residual.dtype == torch.float32):
        if residual is not None and not residual.is_contiguous():
            residual = residual.contiguous()
        if wants_grad:
            return _PointwiseConvFused.apply(x, weight, bias, cloud_bias, residual, relu_in, relu, relu_after)
        cout, cin = weight.shape[:2]
        b = bias
        if cloud_bias is not None:
            b = cloud_bias if bias is None else cloud_bias + bias
        return mfma_linear(x, weight.view(cout, cin), b, relu=relu, residual=residual, x_relu=relu_in, relu_after=relu_after,
                           bias_per_cloud=cloud_bias is not None)
    if cloud_bias is not None:
        cb = cloud_bias if bias is None else cloud_bias + bias
        h = pointwise_conv(torch.relu(x) if relu_in else x, weight, None) + cb.view(cb.shape + (1,) * (x.dim() - 2))
        h = torch.relu_(h) if relu else h
    else:
        h = pointwise_conv(torch.relu(x) if relu_in else x, weight, bias, relu=relu)
    if residual is not None:
        h = h + residual
    return torch.relu(h) if relu_after else h


class _PointwiseConvDual(Function):
    """(W1 x, W2 x) of ONE input as one GEMM with two contiguous outputs (mvp_pointwise_mfma_ex, m_split): a residual unit's
    conv1 / conv_res (vrcnet.py:160-172).  The stacked convolution of round 4 handed out torch.split views -- a copy for the
    consumer that needs a contiguous tensor forward, a concatenation of the two gradients backward."""

    @staticmethod
    def forward(ctx, x, w1, w2):
        c1, c2, cin = w1.size(0), w2.size(0), w1.size(1)
        y1, y2 = mfma_linear(x, torch.cat((w1.view(c1, cin), w2.view(c2, cin)), 0), m_split=c1)
        ctx.save_for_backward(x, w1, w2)
        return y1, y2

    @staticmethod
    def backward(ctx, g1, g2):
        x, w1, w2 = ctx.saved_tensors
        c1, c2, cin = w1.size(0), w2.size(0), w1.size(1)
        g1, g2 = g1.contiguous(), g2.contiguous()
        gx = None
        if ctx.needs_input_grad[0]:
            gx = mfma_linear(g1, w1.view(c1, cin), w_kmajor=True)
            gx = mfma_linear(g2, w2.view(c2, cin), w_kmajor=True, residual=gx)       # W1^T g1 + W2^T g2
        gw1 = mfma_wgrad(x, g1, c1, cin, False)[0].view_as(w1) if ctx.needs_input_grad[1] else None
        gw2 = mfma_wgrad(x, g2, c2, cin, False)[0].view_as(w2) if ctx.needs_input_grad[2] else None
        return gx, gw1, gw2


def pointwise_conv_dual(x, w1, w2):
    """(conv(x, w1), conv(x, w2)), bias-free, both contiguous; one pass over x where the MFMA kernels cover the layer."""
    c1, c2, cin = w1.size(0), w2.size(0), w1.size(1)
    need_x = torch.is_grad_enabled() and x.requires_grad
    stacked_ok = c1 % 32 == 0 and w1.is_contiguous() and w2.is_contiguous() and x.is_cuda \
        and _fused_routes(x, w1, need_x) and _fused_routes(x, w2, need_x) and _mfma_fwd(x, cin, c1 + c2, None)
    if not stacked_ok:
        return pointwise_conv(x, w1), pointwise_conv(x, w2)
    if torch.is_grad_enabled() and (x.requires_grad or w1.requires_grad or w2.requires_grad):
        return _PointwiseConvDual.apply(x, w1, w2)
    return mfma_linear(x, torch.cat((w1.view(c1, cin), w2.view(c2, cin)), 0), m_split=c1)


class _PointwiseConvMax(Function):
    """max over the positions of y = W x + bias -> (B, Cout), with the backward pass the max makes possible:
    grad_y is zero except at the B * Cout winning positions, so the weight gradient is a gather of those
    columns of x and the data gradient a scatter of scaled weight rows -- 2 * B * Cout * Cin multiply-adds
    each instead of two dense GEMMs over all B * L positions on a tensor of zeros (the PointNet stage of
    PCN / VRCNet, 64 x (512 -> 1024) x 2048: 1.16 + 1.40 ms of GEMMs -> the two index passes below).  Values and
    gradients are those of `conv(x).max(dim=-1)[0]` under autograd (the gradient goes to the position torch.max
    reports); tests/test_harness_cpu.py::test_pointwise_conv_max_matches_autograd."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        with torch.no_grad():
            fused = mfma_conv_max(x, weight, bias)
            if fused is not None:
                val, idx = fused
            else:
                y = pointwise_conv(x, weight, bias)
                val, idx = y.flatten(2).max(dim=2)
        ctx.save_for_backward(x, weight, idx)
        ctx.has_bias = bias is not None
        return val

    @staticmethod
    def backward(ctx, g):
        x, weight, idx = ctx.saved_tensors
        B, cin = x.shape[:2]
        cout = weight.size(0)
        x3 = x.reshape(B, cin, -1)
        w2 = weight.reshape(cout, cin)
        g = g.contiguous()
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        length = x3.size(2)
        gx = gw = gb = None
        if _conv_max_kernel_covers(x, weight) and g.dtype == torch.float32 and x3.is_contiguous() and w2.is_contiguous():
            gx = torch.empty_like(x3) if need_x else None
            gw = torch.empty_like(w2) if (need_w or need_b) else None
            gb = torch.empty(cout, dtype=torch.float32, device=x.device) if need_b else None
            nbytes = pointwise_max_backward_scratch_bytes(B, cin, cout, length) if gw is not None else 0
            scratch = _wgrad_scratch(x.device, nbytes) if nbytes else None     # the winning columns of x, staged
            call("mvp_pointwise_max_backward", x.device, B, cin, cout, length, x3, w2, g, idx.int(), gx, gw, gb, scratch,
                 nbytes)
            return (gx.view_as(x) if need_x else None, gw.view_as(weight) if need_w else None, gb)
        # host tensors / other dtypes (the CPU tests): the same two index passes in PyTorch.  CUDA float32 shapes the
        # kernel does not cover never get here (pointwise_conv_max routes them to conv + max under plain autograd: this
        # formulation's expanded index tensors would cost more than the dense GEMMs)
        where = idx.unsqueeze(1).expand(B, cin, cout)                  # [b, ci, co] -> winning position of (b, co)
        if need_x:
            gx = torch.zeros_like(x3).scatter_add_(2, where, g.unsqueeze(1) * w2.t().unsqueeze(0)).view_as(x)
        if need_w:
            gw = torch.einsum("bo,bio->oi", g, torch.gather(x3, 2, where)).view_as(weight)
        if need_b:
            gb = g.sum(0)
        return gx, gw, gb


def mfma_conv_max(x, weight, bias=None):
    """(values (B, Cout), positions (B, Cout) int32) of (W x + bias).max over the positions inside the GEMM's epilogue
    (mvp_pointwise_mfma_max: the (B, Cout, L) tensor is never written), or None where the forward GEMM is not routed to
    the MFMA kernel.  No autograd."""
    cout, cin = weight.size(0), weight.size(1)
    if not (x.is_cuda and weight.dtype == torch.float32 and weight.is_contiguous() and x.numel() > 0
            and _mfma_fwd(x, cin, cout, weight)):
        return None
    B, length = x.size(0), x[0, 0].numel()
    w2d = weight.reshape(cout, cin)
    ldw = 0
    if cin % 4 != 0:
        w2d = F.pad(w2d, (0, -cin % 4))
        ldw = w2d.size(1)
    val = torch.empty(B, cout, dtype=torch.float32, device=x.device)
    idx = torch.empty(B, cout, dtype=torch.int32, device=x.device)
    keys = torch.empty(B * cout, dtype=torch.int64, device=x.device)
    call("mvp_pointwise_mfma_max", x.device, B, cin, cout, length, x, w2d, ldw, bias, 0, val, idx, keys, keys.numel() * 8)
    return val, idx


def _conv_max_kernel_covers(x, weight):
    """Shapes mvp_pointwise_max_backward takes: its per-cloud sort of the winners lives in LDS ((3 Cout + L) * 4 <= 30000
    bytes: L <= 4428 positions at Cout = 1024)."""
    cout = weight.size(0)
    length = x[0, 0].numel() if x.numel() else 0
    return x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and length <= 16384 and cout <= 4096 \
        and (3 * cout + length) * 4 <= 30000 and x.size(0) <= 65535


_conv_max_uncovered_seen = set()


def pointwise_conv_max(x, weight, bias=None):
    """(W x + bias).max over the positions: x (B,Cin,N) / (B,Cin,H,W), weight (Cout,Cin,1[,1]) -> (B, Cout)."""
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
        if not x.is_cuda or _conv_max_kernel_covers(x, weight):
            return _PointwiseConvMax.apply(x, weight, bias)
        shape = (weight.size(0), x[0, 0].numel())
        if shape not in _conv_max_uncovered_seen:          # (once per shape)
            _conv_max_uncovered_seen.add(shape)
            import logging
            logging.getLogger(__name__).info("conv -> max over %d positions x %d channels: outside the sparse backward kernel's "
                                             "LDS budget, dense autograd route", shape[1], shape[0])
        return pointwise_conv(x, weight, bias).flatten(2).max(dim=2)[0]      # (plain autograd)
    fused = mfma_conv_max(x, weight, bias)                                     # no gradient wanted: the fused forward alone
    if fused is not None:
        return fused[0]
    return pointwise_conv(x, weight, bias).flatten(2).max(dim=2)[0]


class PointwiseConv1d(nn.Conv1d):
    """nn.Conv1d(kernel_size=1); `layer(x, relu=True)` = relu(layer(x)) with the
    activation fused into the GEMM's epilogue."""

    def __init__(self, c_in, c_out, bias=True):
        super().__init__(c_in, c_out, kernel_size=1, bias=bias)

    def forward(self, x, relu=False):
        return pointwise_conv(x, self.weight, self.bias, relu=relu)

    def max_over_positions(self, x):
        """self(x).max(dim=2)[0] with the sparse backward pass of _PointwiseConvMax."""
        return pointwise_conv_max(x, self.weight, self.bias)


class PointwiseConv2d(nn.Conv2d):
    def __init__(self, c_in, c_out, bias=True):
        super().__init__(c_in, c_out, kernel_size=1, bias=bias)

    def forward(self, x, relu=False):
        return pointwise_conv(x, self.weight, self.bias, relu=relu)

    def max_over_positions(self, x):
        """self(x).flatten(2).max(dim=2)[0] with the sparse backward pass of _PointwiseConvMax."""
        return pointwise_conv_max(x, self.weight, self.bias)
