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

from model_utils import (EF_expansion, calc_cd, calc_emd, edge_preserve_sampling,
                         furthest_point_sample, gather_points, gen_grid_up, get_edge_features, knn,
                         three_nn_upsampling)
from mm3d_pn2 import three_interpolate
from models.pcn import PCN_encoder


class SA_module(nn.Module):
    """Point self-attention over a fixed kNN graph: relation features of the
    centre and its k neighbours produce per-neighbour weights (shared across
    `share_planes` channel groups) that aggregate the neighbours' values."""

    def __init__(self, in_planes, rel_planes, mid_planes, out_planes, share_planes=8, k=16):
        super(SA_module, self).__init__()
        self.share_planes = share_planes
        self.k = k
        self.conv1 = nn.Conv2d(in_planes, rel_planes, kernel_size=1)
        self.conv2 = nn.Conv2d(in_planes, rel_planes, kernel_size=1)
        self.conv3 = nn.Conv2d(in_planes, mid_planes, kernel_size=1)
        self.conv_w = nn.Sequential(
            nn.ReLU(inplace=False),
            nn.Conv2d(rel_planes * (k + 1), mid_planes // share_planes, kernel_size=1, bias=False),
            nn.ReLU(inplace=False),
            nn.Conv2d(mid_planes // share_planes, k * mid_planes // share_planes, kernel_size=1))
        self.activation_fn = nn.ReLU(inplace=False)
        self.conv_out = nn.Conv2d(mid_planes, out_planes, kernel_size=1)

    def forward(self, input):
        x, idx = input                                   # x: (B, C, 1, N), idx: (B, N, k)
        batch_size, _, _, num_points = x.size()
        act = self.activation_fn(x)
        nbr = get_edge_features(act, idx)                # (B, C, k, N)
        query = self.conv1(act)                          # (B, r, 1, N)
        keys = self.conv2(nbr).reshape(batch_size, -1, 1, num_points)    # (B, k*r, 1, N)
        values = self.conv3(nbr)                         # (B, mid, k, N)

        w = self.conv_w(torch.cat([query, keys], 1)).view(batch_size, -1, self.k, num_points)
        w = w.repeat(1, self.share_planes, 1, 1)         # (B, mid, k, N)
        out = (w * values).sum(dim=2, keepdim=True)
        out = self.conv_out(self.activation_fn(out))     # (B, C_out, 1, N)
        return [out + x, idx]


class Folding(nn.Module):
    """Local folding: every point feature is repeated step_ratio times and
    concatenated with the global feature and a small 2-D grid."""

    def __init__(self, input_size, output_size, step_ratio, global_feature_size=1024, num_models=1):
        super(Folding, self).__init__()
        self.input_size = input_size
        self.output_size = output_size
        self.step_ratio = step_ratio
        self.num_models = num_models
        self.conv = nn.Conv1d(input_size + global_feature_size + 2, output_size, 1, bias=True)
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
        return F.relu(self.conv(features))


class Linear_ResBlock(nn.Module):
    def __init__(self, input_size=1024, output_size=256):
        super(Linear_ResBlock, self).__init__()
        self.conv1 = nn.Linear(input_size, input_size)
        self.conv2 = nn.Linear(input_size, output_size)
        self.conv_res = nn.Linear(input_size, output_size)
        self.af = nn.ReLU(inplace=False)

    def forward(self, feature):
        return self.conv2(self.af(self.conv1(self.af(feature)))) + self.conv_res(feature)


class SK_SA_module(nn.Module):
    """Selective-kernel fusion of several SA_modules with different k."""

    def __init__(self, in_planes, rel_planes, mid_planes, out_planes, share_planes=8, k=[10, 20], r=2, L=32):
        super(SK_SA_module, self).__init__()
        self.num_kernels = len(k)
        d = max(int(out_planes / r), L)
        self.sams = nn.ModuleList(
            [SA_module(in_planes, rel_planes, mid_planes, out_planes, share_planes, kk) for kk in k])
        self.fc = nn.Linear(out_planes, d)
        self.fcs = nn.ModuleList([nn.Linear(d, out_planes) for _ in k])
        self.softmax = nn.Softmax(dim=1)
        self.af = nn.ReLU(inplace=False)

    def forward(self, input):
        x, idxs = input
        assert self.num_kernels == len(idxs)
        feas = torch.stack([self.af(sam([x, idx])[0]) for sam, idx in zip(self.sams, idxs)], dim=1)
        fea_z = self.fc(feas.sum(dim=1).mean(-1).mean(-1))                       # (B, d)
        attention = self.softmax(torch.stack([fc(fea_z) for fc in self.fcs], dim=1))   # (B, K, C)
        fea_v = (feas * attention.unsqueeze(-1).unsqueeze(-1)).sum(dim=1)
        return [fea_v, idxs]


class SKN_Res_unit(nn.Module):
    def __init__(self, input_size, output_size, k=[10, 20], layers=1):
        super(SKN_Res_unit, self).__init__()
        self.conv1 = nn.Conv2d(input_size, output_size, 1, bias=False)
        self.sam = self._make_layer(output_size, output_size // 16, output_size // 4, output_size, int(layers), 8, k=k)
        self.conv2 = nn.Conv2d(output_size, output_size, 1, bias=False)
        self.conv_res = nn.Conv2d(input_size, output_size, 1, bias=False)
        self.af = nn.ReLU(inplace=False)

    def _make_layer(self, in_planes, rel_planes, mid_planes, out_planes, blocks, share_planes=8, k=16):
        return nn.Sequential(*[SK_SA_module(in_planes, rel_planes, mid_planes, out_planes, share_planes, k)
                               for _ in range(blocks)])

    def forward(self, feat, idx):
        x, _ = self.sam([self.conv1(feat), idx])
        return self.conv2(self.af(x)) + self.conv_res(feat)


class SA_SKN_Res_encoder(nn.Module):
    """4-level relational U-Net: SKN residual units on kNN graphs, edge-preserved
    FPS pooling down, three_nn interpolation up."""

    def __init__(self, input_size=3, k=[10, 20], pk=16, output_size=64, layers=[2, 2, 2, 2],
                 pts_num=[3072, 1536, 768, 384]):
        super(SA_SKN_Res_encoder, self).__init__()
        self.init_channel = 64
        c1 = self.init_channel
        c2, c3, c4 = c1 * 2, c1 * 4, c1 * 8
        self.sam_res1 = SKN_Res_unit(input_size, c1, k, int(layers[0]))
        self.sam_res2 = SKN_Res_unit(c2, c2, k, int(layers[1]))
        self.sam_res3 = SKN_Res_unit(c3, c3, k, int(layers[2]))
        self.sam_res4 = SKN_Res_unit(c4, c4, k, int(layers[3]))

        self.conv5 = nn.Conv2d(c4, 1024, 1)
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 1024)

        self.conv6 = nn.Conv2d(c4 + 1024, c4, 1)
        self.conv7 = nn.Conv2d(c3 + c4, c3, 1)
        self.conv8 = nn.Conv2d(c2 + c3, c2, 1)
        self.conv9 = nn.Conv2d(c1 + c2, c1, 1)

        self.conv_out = nn.Conv2d(c1, output_size, 1)
        self.dropout = nn.Dropout()
        self.af = nn.ReLU(inplace=False)
        self.k = k
        self.pk = pk
        self.rate = 2
        self.pts_num = pts_num

    def _graphs(self, pts_bcn):
        """kNN index lists (one per k) of a (B, 3, N) cloud."""
        return [knn(pts_bcn, kk) for kk in self.k]

    def _edge_pooling(self, features, points, rate=2, k=16, sample_num=None):
        features = features.squeeze(2)
        if sample_num is None:
            sample_num = int(features.size(2)) // rate
        ds_features, p_idx, pn_idx, ds_points = edge_preserve_sampling(features.contiguous(), points, sample_num, k)
        return ds_features.unsqueeze(2), p_idx, pn_idx, ds_points

    def _edge_unpooling(self, features, src_pts, tgt_pts):
        idx, weight = three_nn_upsampling(tgt_pts, src_pts)
        return three_interpolate(features.squeeze(2).contiguous(), idx, weight).unsqueeze(2)

    def forward(self, features):
        batch_size = features.size(0)
        xyz = features[:, 0:3, :]
        pts = [xyz.transpose(1, 2).contiguous()]                   # (B, N, 3) per level
        units = [self.sam_res1, self.sam_res2, self.sam_res3, self.sam_res4]

        skips = [self.af(units[0](features.unsqueeze(2), self._graphs(xyz)))]
        for level in range(1, 4):
            x, _, _, p = self._edge_pooling(skips[-1], pts[-1], self.rate, self.pk, self.pts_num[level])
            pts.append(p)
            skips.append(self.af(units[level](x, self._graphs(p.transpose(1, 2).contiguous()))))

        g = self.conv5(skips[3]).max(dim=-1)[0].view(batch_size, -1)
        g = self.dropout(self.af(self.fc2(self.dropout(self.af(self.fc1(g))))))
        g = g.unsqueeze(2).expand(-1, -1, self.pts_num[3]).unsqueeze(2)

        x = self.af(self.conv6(torch.cat([g, skips[3]], 1)))
        for level, conv in ((2, self.conv7), (1, self.conv8), (0, self.conv9)):
            x = self._edge_unpooling(x, pts[level + 1], pts[level])
            x = self.af(conv(torch.cat([x, skips[level]], 1)))
        return self.conv_out(x).squeeze(2)


class MSAP_SKN_decoder(nn.Module):
    """coarse_raw (MLP) -> relational features of (coarse_raw + input) points
    -> coarse_high -> FPS to num_fps -> score-ranked top num_coarse -> fine by
    local folding (or edge-feature expansion)."""

    def __init__(self, num_coarse_raw, num_fps, num_coarse, num_fine, layers=[2, 2, 2, 2], knn_list=[10, 20], pk=10,
                 points_label=False, local_folding=False):
        super(MSAP_SKN_decoder, self).__init__()
        self.num_coarse_raw = num_coarse_raw
        self.num_fps = num_fps
        self.num_coarse = num_coarse
        self.num_fine = num_fine
        self.points_label = points_label
        self.local_folding = local_folding

        self.fc1 = nn.Linear(1024, 1024)
        self.fc2 = nn.Linear(1024, 1024)
        self.fc3 = nn.Linear(1024, num_coarse_raw * 3)

        self.dense_feature_size = 256
        self.expand_feature_size = 64
        self.input_size = 4 if points_label else 3

        self.encoder = SA_SKN_Res_encoder(input_size=self.input_size, k=knn_list, pk=pk,
                                          output_size=self.dense_feature_size, layers=layers)

        self.up_scale = int(math.ceil(num_fine / (num_coarse_raw + 2048)))
        if self.up_scale >= 2:
            self.expansion1 = EF_expansion(input_size=self.dense_feature_size, output_size=self.expand_feature_size,
                                           step_ratio=self.up_scale, k=4)
            self.conv_cup1 = nn.Conv1d(self.expand_feature_size, self.expand_feature_size, 1)
        else:
            self.expansion1 = None
            self.conv_cup1 = nn.Conv1d(self.dense_feature_size, self.expand_feature_size, 1)
        self.conv_cup2 = nn.Conv1d(self.expand_feature_size, 3, 1, bias=True)

        self.conv_s1 = nn.Conv1d(self.expand_feature_size, 16, 1, bias=True)
        self.conv_s2 = nn.Conv1d(16, 8, 1, bias=True)
        self.conv_s3 = nn.Conv1d(8, 1, 1, bias=True)

        ratio = num_fine // num_coarse
        if self.local_folding:
            self.expansion2 = Folding(input_size=self.expand_feature_size, output_size=self.dense_feature_size,
                                      step_ratio=ratio)
        else:
            self.expansion2 = EF_expansion(input_size=self.expand_feature_size, output_size=self.dense_feature_size,
                                           step_ratio=ratio, k=4)
        self.conv_f1 = nn.Conv1d(self.dense_feature_size, self.expand_feature_size, 1)
        self.conv_f2 = nn.Conv1d(self.expand_feature_size, 3, 1)
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

        coarse_features = self.af(self.conv_cup1(dense_feat))
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
                fine = self.conv_f2(self.af(self.conv_f1(up_features))) + center
            else:
                fine = self.conv_f2(self.af(self.conv_f1(self.expansion2(coarse_features))))
        else:
            assert coarse.size(2) == self.num_fine
            fine = coarse
        return coarse_raw, coarse_high, coarse, fine


class Model(nn.Module):
    def __init__(self, args, size_z=128, global_feature_size=1024):
        super(Model, self).__init__()
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
            emd = calc_emd(fine, gt, eps=0.004, iterations=3000) if self.eval_emd else 0
            cd_p, cd_t, f1 = calc_cd(fine, gt, calc_f1=True)
            return {'out1': coarse_raw, 'out2': fine, 'emd': emd, 'cd_p': cd_p, 'cd_t': cd_t, 'f1': f1}
        return {'result': fine}
