"""VRCNet (Variational Relational point Completion Network) -- counterpart of
the reference's completion/models/vrcnet.py (SA_module :21-57, Folding :60-91,
Linear_ResBlock :94-104, SK_SA_module :107-152, SKN_Res_unit :155-173,
SA_SKN_Res_encoder :176-298, MSAP_SKN_decoder :301-411, Model :414-526).

Same sub-module / parameter names (checkpoints interchange) and the same
forward contract as PCN/ECG.  Op-layer calls per forward at the default cfg
(training doubles the batch, :452-454): FPS gt 2048->2048, FPS
3072->1536->768->384 with gather/group inside edge_preserve_sampling,
three_nn + three_interpolate 384->768->1536->3072, FPS 3072->2048 + gathers,
score-ranked gathers, and 4 Chamfer distances.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from model_utils import EF_expansion, calc_cd, furthest_point_sample, gather_points, gen_grid_up
from models._common import dense, eval_outputs, pointwise1d
from models.pcn import PCN_encoder
from models.relational import SA_module, SA_SKN_Res_encoder, SK_SA_module, SKN_Res_unit  # noqa: F401


class Folding(nn.Module):
    """Local folding: every point feature is repeated step_ratio times and
    concatenated with the global feature and a small 2-D grid."""

    def __init__(self, input_size, output_size, step_ratio, global_feature_size=1024, num_models=1):
        super().__init__()
        self.input_size = input_size
        self.output_size = output_size
        self.step_ratio = step_ratio
        self.num_models = num_models
        self.conv = pointwise1d(input_size + global_feature_size + 2, output_size, bias=True)
        # (step_ratio, 2) grid over [-0.2, 0.2]^2; plain attribute as in the
        # reference (not part of the checkpoint)
        self.grid = gen_grid_up(step_ratio, 0.2).transpose(0, 1).contiguous()

    def forward(self, point_feat, global_feat):
        batch_size, num_features, num_points = point_feat.size()
        total = num_points * self.step_ratio
        point_feat = point_feat.unsqueeze(3).expand(-1, -1, -1, self.step_ratio).reshape(batch_size, num_features, total)
        global_feat = global_feat.unsqueeze(2).expand(-1, -1, total).repeat(self.num_models, 1, 1)
        grid_feat = self.grid.to(point_feat.device).unsqueeze(0).repeat(batch_size, num_points, 1).transpose(1, 2)
        features = torch.cat([global_feat, point_feat, grid_feat], dim=1)
        return self.conv(features.contiguous(), relu=True)


class Linear_ResBlock(nn.Module):
    def __init__(self, input_size=1024, output_size=256):
        super().__init__()
        self.conv1 = dense(input_size, input_size)
        self.conv2 = dense(input_size, output_size)
        self.conv_res = dense(input_size, output_size)
        self.af = nn.ReLU(inplace=False)

    def forward(self, feature):
        return self.conv2(self.af(self.conv1(self.af(feature)))) + self.conv_res(feature)


class MSAP_SKN_decoder(nn.Module):
    """coarse_raw (MLP) -> relational features of (coarse_raw + input) points
    -> coarse_high -> FPS to num_fps -> score-ranked top num_coarse -> fine by
    local folding (or edge-feature expansion)."""

    def __init__(self, num_coarse_raw, num_fps, num_coarse, num_fine, layers=[2, 2, 2, 2], knn_list=[10, 20], pk=10,
                 points_label=False, local_folding=False):
        super().__init__()
        self.num_coarse_raw = num_coarse_raw
        self.num_fps = num_fps
        self.num_coarse = num_coarse
        self.num_fine = num_fine
        self.points_label = points_label
        self.local_folding = local_folding

        self.fc1 = dense(1024, 1024)
        self.fc2 = dense(1024, 1024)
        self.fc3 = dense(1024, num_coarse_raw * 3)

        self.dense_feature_size = 256
        self.expand_feature_size = 64
        self.input_size = 4 if points_label else 3

        self.encoder = SA_SKN_Res_encoder(input_size=self.input_size, k=knn_list, pk=pk,
                                          output_size=self.dense_feature_size, layers=layers)

        self.up_scale = int(math.ceil(num_fine / (num_coarse_raw + 2048)))
        if self.up_scale >= 2:
            self.expansion1 = EF_expansion(input_size=self.dense_feature_size, output_size=self.expand_feature_size,
                                           step_ratio=self.up_scale, k=4)
            self.conv_cup1 = pointwise1d(self.expand_feature_size, self.expand_feature_size)
        else:
            self.expansion1 = None
            self.conv_cup1 = pointwise1d(self.dense_feature_size, self.expand_feature_size)
        self.conv_cup2 = pointwise1d(self.expand_feature_size, 3, bias=True)

        self.conv_s1 = pointwise1d(self.expand_feature_size, 16, bias=True)
        self.conv_s2 = pointwise1d(16, 8, bias=True)
        self.conv_s3 = pointwise1d(8, 1, bias=True)

        ratio = num_fine // num_coarse
        if self.local_folding:
            self.expansion2 = Folding(input_size=self.expand_feature_size, output_size=self.dense_feature_size,
                                      step_ratio=ratio)
        else:
            self.expansion2 = EF_expansion(input_size=self.expand_feature_size, output_size=self.dense_feature_size,
                                           step_ratio=ratio, k=4)
        self.conv_f1 = pointwise1d(self.dense_feature_size, self.expand_feature_size)
        self.conv_f2 = pointwise1d(self.expand_feature_size, 3)
        self.af = nn.ReLU(inplace=False)

    def forward(self, global_feat, point_input):
        batch_size = global_feat.size(0)
        coarse_raw = self.fc3(self.af(self.fc2(self.af(self.fc1(global_feat))))) \
            .view(batch_size, 3, self.num_coarse_raw)

        if self.points_label:      # 4th channel tells generated (0) from observed (1) points
            zeros = coarse_raw.new_zeros(batch_size, 1, coarse_raw.shape[2])
            ones = point_input.new_ones(batch_size, 1, point_input.shape[2])
            points = torch.cat((torch.cat((coarse_raw, zeros), 1), torch.cat((point_input, ones), 1)), 2)
        else:
            points = torch.cat((coarse_raw, point_input), 2)
        dense_feat = self.encoder(points)
        if self.up_scale >= 2:
            dense_feat = self.expansion1(dense_feat)

        coarse_features = self.conv_cup1(dense_feat, relu=True)
        coarse_high = self.conv_cup2(coarse_features)

        if coarse_high.size(2) > self.num_fps:
            idx_fps = furthest_point_sample(coarse_high.transpose(1, 2).contiguous(), self.num_fps)
            coarse_fps = gather_points(coarse_high.contiguous(), idx_fps)
            coarse_features = gather_points(coarse_features.contiguous(), idx_fps)
        else:
            coarse_fps = coarse_high

        if coarse_fps.size(2) > self.num_coarse:
            scores = F.softplus(self.conv_s3(self.af(self.conv_s2(self.af(self.conv_s1(coarse_features))))))
            idx_scores = scores.topk(k=self.num_coarse, dim=2)[1].view(batch_size, -1).int()
            coarse = gather_points(coarse_fps.contiguous(), idx_scores)
            coarse_features = gather_points(coarse_features.contiguous(), idx_scores)
        else:
            coarse = coarse_fps

        if coarse.size(2) < self.num_fine:
            if self.local_folding:
                up_features = self.expansion2(coarse_features, global_feat)
                ratio = self.num_fine // self.num_coarse
                center = coarse.unsqueeze(3).expand(-1, -1, -1, ratio).reshape(batch_size, 3, self.num_fine)
                fine = self.conv_f2(self.conv_f1(up_features, relu=True)) + center
            else:
                fine = self.conv_f2(self.conv_f1(self.expansion2(coarse_features), relu=True))
        else:
            assert coarse.size(2) == self.num_fine
            fine = coarse
        return coarse_raw, coarse_high, coarse, fine


class Model(nn.Module):
    """Variational completion network.  Training runs two paths through ONE doubled batch: the
    reconstruction path (posterior q(z | partial) against the prior p(z | complete)) and the completion
    path; both decode `feat_partial + generator(z)`.  Outputs of forward():
      train: (fine (2B, num_points, 3), per-cloud CD of fine (2B,), scalar loss =
              10 CD(coarse_raw) + 0.5 CD(coarse_high) + CD(coarse) + alpha CD(fine) + 20 (KLD or MMD terms))
      val:   metric dictionary of models/_common.eval_outputs;   test: {'result': fine}
    `args`: layers, knn_list, pk, local_folding, points_label, num_coarse_raw, num_fps, num_coarse,
    num_points, distribution_loss ('KLD' | 'MMD'), loss ('cd'), eval_emd."""

    def __init__(self, args, size_z=128, global_feature_size=1024):
        super().__init__()
        layers = [int(i) for i in str(args.layers).split(',')]
        knn_list = [int(i) for i in str(args.knn_list).split(',')]

        self.size_z = size_z
        self.distribution_loss = args.distribution_loss
        self.train_loss = args.loss
        self.eval_emd = args.eval_emd
        self.encoder = PCN_encoder(output_size=global_feature_size)
        self.posterior_infer1 = Linear_ResBlock(input_size=global_feature_size, output_size=global_feature_size)
        self.posterior_infer2 = Linear_ResBlock(input_size=global_feature_size, output_size=size_z * 2)
        self.prior_infer = Linear_ResBlock(input_size=global_feature_size, output_size=size_z * 2)
        self.generator = Linear_ResBlock(input_size=size_z, output_size=global_feature_size)
        self.decoder = MSAP_SKN_decoder(num_fps=args.num_fps, num_fine=args.num_points, num_coarse=args.num_coarse,
                                        num_coarse_raw=args.num_coarse_raw, layers=layers, knn_list=knn_list,
                                        pk=args.pk, local_folding=args.local_folding, points_label=args.points_label)

    @staticmethod
    def compute_kernel(x, y):
        dim = x.size(1)
        diff = x.unsqueeze(1) - y.unsqueeze(0)
        return torch.exp(-(diff ** 2).mean(dim=2) / float(dim))

    def mmd_loss(self, x, y):
        return self.compute_kernel(x, x).mean() + self.compute_kernel(y, y).mean() \
            - 2 * self.compute_kernel(x, y).mean()

    mmd_loss2 = mmd_loss   # the reference calls an undefined `mmd_loss2` (:498); same estimator

    def _normal(self, raw):
        mu, std = torch.split(raw, self.size_z, dim=1)
        return mu, F.softplus(std)

    def forward(self, x, gt=None, prefix="train", mean_feature=None, alpha=None):
        num_input = x.size(2)
        train = prefix == "train"

        if train:
            # reconstruction path sees the (sub-sampled) complete shape; both
            # paths are decoded in one doubled batch (:450-455)
            y = gather_points(gt.transpose(1, 2).contiguous(), furthest_point_sample(gt, num_input))
            gt = torch.cat([gt, gt], dim=0)
            feat = self.encoder(torch.cat([x, y], dim=0))
            x = torch.cat([x, x], dim=0)
            feat_x, feat_y = feat.chunk(2)
            q_mu, q_std = self._normal(self.posterior_infer2(self.posterior_infer1(feat_x)))
            p_mu, p_std = self._normal(self.prior_infer(feat_y))
            q_distribution = torch.distributions.Normal(q_mu, q_std)
            p_distribution = torch.distributions.Normal(p_mu, p_std)
            p_distribution_fix = torch.distributions.Normal(p_mu.detach(), p_std.detach())
            m_distribution = torch.distributions.Normal(torch.zeros_like(p_mu), torch.ones_like(p_std))
            z = torch.cat([q_distribution.rsample(), p_distribution.rsample()], dim=0)
            feat = torch.cat([feat_x, feat_x], dim=0)
        else:
            feat = self.encoder(x)
            q_mu, q_std = self._normal(self.posterior_infer2(self.posterior_infer1(feat)))
            q_distribution = torch.distributions.Normal(q_mu, q_std)
            z = q_distribution.rsample()

        feat = feat + self.generator(z)
        coarse_raw, coarse_high, coarse, fine = [t.transpose(1, 2).contiguous() for t in self.decoder(feat, x)]

        if train:
            if self.distribution_loss == 'MMD':
                z_m, z_q = m_distribution.rsample(), q_distribution.rsample()
                z_p, z_p_fix = p_distribution.rsample(), p_distribution_fix.rsample()
                dl_rec = self.mmd_loss(z_m, z_p)
                dl_g = self.mmd_loss2(z_q, z_p_fix)
            elif self.distribution_loss == 'KLD':
                dl_rec = torch.distributions.kl_divergence(m_distribution, p_distribution)
                dl_g = torch.distributions.kl_divergence(p_distribution_fix, q_distribution)
            else:
                raise NotImplementedError('Distribution loss is either MMD or KLD')
            if self.train_loss != 'cd':
                raise NotImplementedError('Only CD is supported')
            loss1, loss2, loss3, loss4 = [calc_cd(o, gt)[0] for o in (coarse_raw, coarse_high, coarse, fine)]
            total_train_loss = loss1.mean() * 10 + loss2.mean() * 0.5 + loss3.mean() + loss4.mean() * alpha
            total_train_loss = total_train_loss + (dl_rec.mean() + dl_g.mean()) * 20
            return fine, loss4, total_train_loss
        if prefix == "val":
            return eval_outputs(coarse_raw, fine, gt, self.eval_emd)
        return {'result': fine}
