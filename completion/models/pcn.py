"""PCN (Point Completion Network) -- counterpart of the reference's
completion/models/pcn.py (PCN_encoder :13-31, PCN_decoder :34-72, Model
:75-112).  Same sub-module / parameter names (so reference checkpoints load
into it), same forward(x, gt, prefix, mean_feature, alpha) contract:
  prefix="train" -> (fine, loss_fine (B,), total scalar loss)
  prefix="val"   -> {'out1','out2','emd','cd_p','cd_t','f1'}
  otherwise      -> {'result': fine}
The layers are plain PyTorch (rocBLAS/MIOpen under PyTorch-ROCm); the losses
and metrics call the MI355X op layer through model_utils.calc_cd / calc_emd.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from model_utils import gen_grid_up
from models._common import conv_folded_concat, conv_global_concat, dense, eval_outputs, pointwise1d, shape_loss


class PCN_encoder(nn.Module):
    """Two stacked PointNet stages: per-point MLP -> global max -> concat ->
    per-point MLP -> global max."""

    # per-point maps and max-pools only: the feature does not depend on the order of the input points (VRCNet's
    # training path relies on it to skip a full-size FPS that only permutes gt: models/vrcnet.py)
    order_invariant = True

    def __init__(self, output_size=1024):
        super().__init__()
        self.conv1 = pointwise1d(3, 128)
        self.conv2 = pointwise1d(128, 256)
        self.conv3 = pointwise1d(512, 512)
        self.conv4 = pointwise1d(512, output_size)

    def forward(self, x):
        # x (B, 3, N) -> per-point features (B, 256, N) -> global max (B, 256, 1), tiled back and
        # concatenated (B, 512, N) -> second per-point MLP (B, output_size, N) -> global max (B, output_size)
        local = self.conv2(self.conv1(x, relu=True))
        pooled = local.max(dim=2)[0]                                   # (B, 256)
        # conv3 over cat(local, pooled tiled N times) (the reference, pcn.py:25-29) = W[:, :256] local + one vector
        # per cloud (W[:, 256:] pooled + bias): half the reduction, no (B, 512, N) concatenation; same parameters,
        # same function up to float32 summation order (test_pcn_encoder_split_conv3_equals_concatenated_formulation)
        h = conv_global_concat(self.conv3, pooled, local, relu=True, global_first=False)
        return self.conv4.max_over_positions(h)           # conv4(h).max(dim=2)[0], sparse backward (pointwise.py)


class PCN_decoder(nn.Module):
    """Coarse cloud from an MLP, then a folding stage that lifts a small 2-D
    grid around every coarse point."""

    def __init__(self, num_coarse, num_fine, scale, cat_feature_num):
        super().__init__()
        self.num_coarse = num_coarse
        self.num_fine = num_fine
        self.scale = scale
        self.fc1 = dense(1024, 1024)
        self.fc2 = dense(1024, 1024)
        self.fc3 = dense(1024, num_coarse * 3)
        # (2, scale) folding grid; not part of the checkpoint (the reference
        # keeps it as a plain attribute)
        self.register_buffer("grid", gen_grid_up(2 ** (int(math.log2(scale))), 0.05).contiguous(),
                             persistent=False)
        self.conv1 = pointwise1d(cat_feature_num, 512)
        self.conv2 = pointwise1d(512, 512)
        self.conv3 = pointwise1d(512, 3)

    def _folded_conv1(self, x, coarse):
        """relu(conv1(cat(grid_feat, center, global_feat))) without the (B, 1029, Nf) tensor or its GEMM.
        The reference (pcn.py:60-68) tiles the S-point grid under every coarse point, repeats every coarse
        point S times and the global feature Nf times, concatenates and convolves: 2 * 1029 * 512 flops
        per fine point.  conv1 is linear, so with W = [Wg | Wc | Wx] split at channels 2 and 5
            conv1(feat)[b, :, c S + s] = Wg grid[:, s] + Wc coarse[b, :, c] + (Wx x[b] + bias)
        -- a (512 x S) patch shared by everything, one vector per coarse point, one per cloud: three tiny
        products and one broadcast sum.  Same parameters, same function up to float32 summation order
        (tests/test_harness_cpu.py::test_pcn_folded_conv1_equals_concatenated_formulation); at the eval
        setting (32 clouds x 16384 points) it replaces a 4.8 ms GEMM by a 0.5 ms elementwise pass."""
        return conv_folded_concat(self.conv1, [('grid', self.grid.detach()), ('point', coarse), ('global', x)], self.scale)

    def forward(self, x):
        # x: global feature (B, 1024).  Shapes below: S = scale, Nc = num_coarse, Nf = num_fine = Nc * S.
        #   coarse      (B, 3, Nc)     three-layer MLP, reshaped
        #   center      (B, 3, Nf)     every coarse point repeated S times (the fine points fold around it)
        #   grid_feat   (B, 2, Nf)     the same S-point 2-D patch under every coarse point
        #   global_feat (B, 1024, Nf)  the global feature under every fine point
        #   fine        (B, 3, Nf)     per-point MLP on the 2 + 3 + 1024 channels, added to its centre
        batch_size = x.size(0)
        coarse = self.fc3(F.relu(self.fc2(F.relu(self.fc1(x))))).view(-1, 3, self.num_coarse)

        # every coarse point repeated `scale` times: (B, 3, num_fine)
        center = coarse.unsqueeze(3).expand(-1, -1, -1, self.scale).reshape(batch_size, 3, self.num_fine)
        fine = self.conv3(self.conv2(self._folded_conv1(x, coarse), relu=True)) + center
        return coarse, fine


class Model(nn.Module):
    """PCN = PCN_encoder + PCN_decoder + the loss / metric tail shared by the three networks
    (models/_common.py).  `args` needs num_points (fine size), loss ('cd' | 'emd') and eval_emd."""

    def __init__(self, args, num_coarse=1024):
        super().__init__()
        self.num_coarse = num_coarse
        self.num_points = args.num_points
        self.train_loss = args.loss
        self.eval_emd = args.eval_emd
        self.scale = self.num_points // num_coarse
        self.cat_feature_num = 2 + 3 + 1024

        self.encoder = PCN_encoder()
        self.decoder = PCN_decoder(num_coarse, self.num_points, self.scale, self.cat_feature_num)

    def forward(self, x, gt=None, prefix="train", mean_feature=None, alpha=None):
        out1, out2 = self.decoder(self.encoder(x))
        out1 = out1.transpose(1, 2).contiguous()
        out2 = out2.transpose(1, 2).contiguous()

        if prefix == "train":
            loss_coarse = shape_loss(self.train_loss, out1, gt)
            loss_fine = shape_loss(self.train_loss, out2, gt)
            return out2, loss_fine, loss_coarse.mean() + loss_fine.mean() * alpha
        if prefix == "val":
            return eval_outputs(out1, out2, gt, self.eval_emd)
        return {'result': out2}
