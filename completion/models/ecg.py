"""ECG (Edge-aware Completion with Graph convolution) -- counterpart of the
reference's completion/models/ecg.py (Stack_conv :21-33, Dense_conv :36-65,
EF_encoder :68-158, ECG_decoder :161-213, Model :216-253).

Same sub-module / parameter names (checkpoints interchange) and the same
forward(x, gt, prefix, mean_feature, alpha) contract as PCN.  The graph /
sampling structure runs on the MI355X op layer:
  edge_preserve_sampling  -> furthest_point_sample, gather_points, grouping_operation
  three_nn_upsampling     -> three_nn;   three_interpolate
  get_uniform_loss        -> furthest_point_sample, gather_points, ball_query, grouping_operation
  final resampling        -> furthest_point_sample, gather_points
Op shapes at the default cfg (B, 3072 = 1024 coarse + 2048 input points):
FPS 3072->1024->256->64; three_nn/interpolate 64->256 (C=1024), 256->1024
(C=768), 1024->3072 (C=512).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from model_utils import EF_expansion, GeometryAhead, furthest_point_sample, gather_points, get_uniform_loss
from op_config import OPS
from models._common import dense, eval_outputs, pointwise1d, shape_loss
from models.edge_unet import Dense_conv, EF_encoder, Stack_conv  # noqa: F401
from models.pcn import PCN_encoder


class ECG_decoder(nn.Module):
    """Coarse skeleton from the global feature; then per-point features of
    (coarse + input) points, optional feature expansion, and a point head."""

    def __init__(self, num_coarse, num_fine, num_input):
        super().__init__()
        self.num_coarse = num_coarse
        self.num_fine = num_fine
        self.scale = int(math.ceil(num_fine / (num_coarse + num_input)))

        self.fc1 = dense(1024, 1024)
        self.fc2 = dense(1024, 1024)
        self.fc3 = dense(1024, num_coarse * 3)

        self.dense_feature_size = 256
        self.expand_feature_size = 64
        self.input_size = 3

        self.encoder = EF_encoder(growth_rate=24, dense_n=3, k=16, hierarchy=[1024, 256, 64],
                                  input_size=self.input_size, output_size=self.dense_feature_size)
        if self.scale >= 2:
            self.expansion = EF_expansion(input_size=self.dense_feature_size,
                                          output_size=self.expand_feature_size, step_ratio=self.scale, k=4)
            self.conv1 = pointwise1d(self.expand_feature_size, self.expand_feature_size)
        else:
            self.expansion = None
            self.conv1 = pointwise1d(self.dense_feature_size, self.expand_feature_size)
        self.conv2 = pointwise1d(self.expand_feature_size, 3)

    def forward(self, global_feat, point_input, meanwhile=None):
        """meanwhile(coarse): work of the caller that needs the skeleton only (its training losses); issued on the main
        stream while the final FPS -- 2047 sequential rounds on an eighth of the CUs, non-differentiable -- runs on a side lane
        (model_utils.GeometryAhead; op_config fps_beside_losses).  Same launches, same values."""
        batch_size = global_feat.size(0)
        coarse = self.fc3(F.relu(self.fc2(F.relu(self.fc1(global_feat))))).view(batch_size, 3, self.num_coarse)

        dense_feat = self.encoder(torch.cat((coarse, point_input), 2))
        if self.scale >= 2:
            dense_feat = self.expansion(dense_feat)
        fine = self.conv2(F.relu(self.conv1(dense_feat)))

        if fine.size(2) > self.num_fine:      # thin out to exactly num_fine points
            src = fine.transpose(1, 2).contiguous()
            if meanwhile is not None and src.is_cuda and OPS.side_lanes > 0 and OPS.fps_beside_losses:
                geo = GeometryAhead(src.device)
                geo.run("fps", lambda: furthest_point_sample(src.detach(), self.num_fine))
                meanwhile(coarse)
                meanwhile = None
                keep = geo.take("fps")
                geo.join()
            else:
                keep = furthest_point_sample(src, self.num_fine)
            fine = gather_points(fine.contiguous(), keep)
        if meanwhile is not None:
            meanwhile(coarse)
        return coarse, fine


class Model(nn.Module):
    def __init__(self, args, num_coarse=1024, num_input=2048):
        super().__init__()
        self.num_coarse = num_coarse
        self.num_points = args.num_points
        self.train_loss = args.loss
        self.eval_emd = args.eval_emd
        self.encoder = PCN_encoder()
        self.decoder = ECG_decoder(num_coarse, self.num_points, num_input)

    def forward(self, x, gt=None, prefix="train", mean_feature=None, alpha=None):
        if mean_feature:
            raise NotImplementedError
        early = {}

        def skeleton_terms(coarse):          # (needs the skeleton only: issued beside the decoder's final FPS)
            c = coarse.transpose(1, 2).contiguous()
            early['term'] = shape_loss(self.train_loss, c, gt).mean() + 0.1 * get_uniform_loss(c).mean()
        out1, out2 = self.decoder(self.encoder(x), x, meanwhile=skeleton_terms if prefix == "train" else None)
        out1 = out1.transpose(1, 2).contiguous()
        out2 = out2.transpose(1, 2).contiguous()

        if prefix == "train":
            # reconstruction + 0.1 x uniformity, on the skeleton and (x alpha) the fine cloud
            loss_fine = shape_loss(self.train_loss, out2, gt)
            term_coarse = early['term']
            term_fine = loss_fine.mean() + 0.1 * get_uniform_loss(out2).mean()
            return out2, loss_fine, term_coarse + term_fine * alpha
        if prefix == "val":
            return eval_outputs(out1, out2, gt, self.eval_emd)
        return {'result': out2}
