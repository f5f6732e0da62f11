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

from model_utils import (EF_expansion, calc_cd, calc_emd, edge_preserve_sampling,
                         furthest_point_sample, gather_points, get_graph_feature,
                         get_uniform_loss, three_nn_upsampling)
from mm3d_pn2 import three_interpolate
from models.pcn import PCN_encoder


class Stack_conv(nn.Module):
    """1x1 conv whose output is concatenated BEHIND its input (dense growth)."""

    def __init__(self, input_size, output_size, act=None):
        super(Stack_conv, self).__init__()
        self.model = nn.Sequential()
        self.model.add_module('conv', nn.Conv2d(input_size, output_size, 1))
        if act is not None:
            self.model.add_module('act', act)

    def forward(self, x):
        return torch.cat((x, self.model(x)), 1)


class Dense_conv(nn.Module):
    """DenseNet-style edge convolution: kNN edge features -> first_conv ->
    (dense_n - 1) Stack_conv layers -> max over the k neighbours.
    Output channels: input_size + growth_rate * dense_n."""

    def __init__(self, input_size, growth_rate=64, dense_n=3, k=16):
        super(Dense_conv, self).__init__()
        self.growth_rate = growth_rate
        self.dense_n = dense_n
        self.k = k
        self.comp = growth_rate * 2
        self.input_size = input_size

        self.first_conv = nn.Conv2d(input_size * 2, growth_rate, 1)
        width = input_size + growth_rate
        self.model = nn.Sequential()
        for i in range(1, dense_n):
            last = i == dense_n - 1
            self.model.add_module('stack_conv_%d' % i,
                                  Stack_conv(width, growth_rate, None if last else nn.ReLU()))
            width += growth_rate
        self.input_size = width - growth_rate if dense_n > 1 else width

    def forward(self, x):
        edge = F.relu(self.first_conv(get_graph_feature(x, k=self.k)))        # (B, g, N, k)
        edge = torch.cat((edge, x.unsqueeze(3).expand(-1, -1, -1, self.k)), 1)
        return self.model(edge).max(dim=3)[0]


class EF_encoder(nn.Module):
    """4-level edge-feature U-Net over the point cloud: dense edge convs on
    the way down (edge-preserved FPS pooling), three_nn interpolation on the
    way up, skip connections at every level."""

    def __init__(self, growth_rate=24, dense_n=3, k=16, hierarchy=[1024, 256, 64], input_size=3, output_size=256):
        super(EF_encoder, self).__init__()
        self.growth_rate = growth_rate
        self.comp = growth_rate * 2
        self.dense_n = dense_n
        self.k = k
        self.hierarchy = hierarchy
        self.init_channel = 24
        grow = growth_rate * dense_n

        self.conv1 = nn.Conv1d(input_size, self.init_channel, 1)
        self.dense_conv1 = Dense_conv(self.init_channel, growth_rate, dense_n, k)
        c1 = self.init_channel * 2 + grow                      # 120

        self.conv2 = nn.Conv1d(c1 * 2, self.comp, 1)
        self.dense_conv2 = Dense_conv(self.comp, growth_rate, dense_n, k)
        c2 = c1 * 2 + self.comp + grow                         # 360

        self.conv3 = nn.Conv1d(c2 * 2, self.comp, 1)
        self.dense_conv3 = Dense_conv(self.comp, growth_rate, dense_n, k)
        c3 = c2 * 2 + self.comp + grow                         # 840

        self.conv4 = nn.Conv1d(c3 * 2, self.comp, 1)
        self.dense_conv4 = Dense_conv(self.comp, growth_rate, dense_n, k)
        c4 = c3 * 2 + self.comp + grow                         # 1800

        self.gf_conv = nn.Conv1d(c4, 1024, 1)
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 1024)

        self.conv5 = nn.Conv1d(c4 + 1024, 1024, 1)
        self.conv6 = nn.Conv1d(c3 + 1024, 768, 1)
        self.conv7 = nn.Conv1d(c2 + 768, 512, 1)
        self.conv8 = nn.Conv1d(c1 + 512, output_size, 1)

    def forward(self, x):
        pts = [x[:, 0:3, :].transpose(1, 2).contiguous()]       # level-0 coordinates (B,N,3)

        # ---- down: level features f[l] (before pooling), pooled inputs
        x0 = F.relu(self.conv1(x))
        f = [torch.cat((F.relu(self.dense_conv1(x0)), x0), 1)]  # 120 channels
        squeeze = [self.conv2, self.conv3, self.conv4]
        dense = [self.dense_conv2, self.dense_conv3, self.dense_conv4]
        for level in range(3):
            pooled, _, _, p_next = edge_preserve_sampling(f[level], pts[level], self.hierarchy[level], self.k)
            pts.append(p_next)
            y = F.relu(dense[level](F.relu(squeeze[level](pooled))))
            f.append(torch.cat((y, pooled), 1))

        # ---- bottleneck: global feature broadcast back onto the coarsest level
        g = self.gf_conv(f[3]).max(dim=-1)[0]
        g = F.relu(self.fc2(F.relu(self.fc1(g)))).unsqueeze(2).expand(-1, -1, self.hierarchy[2])
        up = F.relu(self.conv5(torch.cat((g, f[3]), 1)))

        # ---- up: interpolate to the finer level, fuse with its skip features
        for level, conv in ((2, self.conv6), (1, self.conv7)):
            idx, weight = three_nn_upsampling(pts[level], pts[level + 1])
            up = three_interpolate(up.contiguous(), idx, weight)
            up = F.relu(conv(torch.cat((f[level], up), 1)))
        idx, weight = three_nn_upsampling(pts[0], pts[1])
        up = three_interpolate(up.contiguous(), idx, weight)
        return self.conv8(torch.cat((f[0], up), 1))


class ECG_decoder(nn.Module):
    """Coarse skeleton from the global feature; then per-point features of
    (coarse + input) points, optional feature expansion, and a point head."""

    def __init__(self, num_coarse, num_fine, num_input):
        super(ECG_decoder, self).__init__()
        self.num_coarse = num_coarse
        self.num_fine = num_fine
        self.scale = int(math.ceil(num_fine / (num_coarse + num_input)))

        self.fc1 = nn.Linear(1024, 1024)
        self.fc2 = nn.Linear(1024, 1024)
        self.fc3 = nn.Linear(1024, num_coarse * 3)

        self.dense_feature_size = 256
        self.expand_feature_size = 64
        self.input_size = 3

        self.encoder = EF_encoder(growth_rate=24, dense_n=3, k=16, hierarchy=[1024, 256, 64],
                                  input_size=self.input_size, output_size=self.dense_feature_size)
        if self.scale >= 2:
            self.expansion = EF_expansion(input_size=self.dense_feature_size,
                                          output_size=self.expand_feature_size, step_ratio=self.scale, k=4)
            self.conv1 = nn.Conv1d(self.expand_feature_size, self.expand_feature_size, 1)
        else:
            self.expansion = None
            self.conv1 = nn.Conv1d(self.dense_feature_size, self.expand_feature_size, 1)
        self.conv2 = nn.Conv1d(self.expand_feature_size, 3, 1)

    def forward(self, global_feat, point_input):
        batch_size = global_feat.size(0)
        coarse = self.fc3(F.relu(self.fc2(F.relu(self.fc1(global_feat))))).view(batch_size, 3, self.num_coarse)

        dense_feat = self.encoder(torch.cat((coarse, point_input), 2))
        if self.scale >= 2:
            dense_feat = self.expansion(dense_feat)
        fine = self.conv2(F.relu(self.conv1(dense_feat)))

        if fine.size(2) > self.num_fine:      # thin out to exactly num_fine points
            keep = furthest_point_sample(fine.transpose(1, 2).contiguous(), self.num_fine)
            fine = gather_points(fine.contiguous(), keep)
        return coarse, fine


class Model(nn.Module):
    def __init__(self, args, num_coarse=1024, num_input=2048):
        super(Model, self).__init__()
        self.num_coarse = num_coarse
        self.num_points = args.num_points
        self.train_loss = args.loss
        self.eval_emd = args.eval_emd
        self.encoder = PCN_encoder()
        self.decoder = ECG_decoder(num_coarse, self.num_points, num_input)

    def forward(self, x, gt=None, prefix="train", mean_feature=None, alpha=None):
        if mean_feature:
            raise NotImplementedError
        out1, out2 = self.decoder(self.encoder(x), x)
        out1 = out1.transpose(1, 2).contiguous()
        out2 = out2.transpose(1, 2).contiguous()

        if prefix == "train":
            uniform_loss1 = get_uniform_loss(out1)
            uniform_loss2 = get_uniform_loss(out2)
            if self.train_loss == 'emd':
                loss1, loss2 = calc_emd(out1, gt), calc_emd(out2, gt)
            elif self.train_loss == 'cd':
                loss1, loss2 = calc_cd(out1, gt)[0], calc_cd(out2, gt)[0]
            else:
                raise NotImplementedError('Train loss is either CD or EMD!')
            total_train_loss = loss1.mean() + uniform_loss1.mean() * 0.1 + \
                (loss2.mean() + uniform_loss2.mean() * 0.1) * alpha
            return out2, loss2, total_train_loss
        if prefix == "val":
            emd = calc_emd(out2, gt, eps=0.004, iterations=3000) if self.eval_emd else 0
            cd_p, cd_t, f1 = calc_cd(out2, gt, calc_f1=True)
            return {'out1': out1, 'out2': out2, 'emd': emd, 'cd_p': cd_p, 'cd_t': cd_t, 'f1': f1}
        return {'result': out2}
