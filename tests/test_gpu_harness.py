"""GPU tests of the callers either side of the op layer (SURVEY 8a rows
F1-F3, G1): model_utils compositions checked against the same composition of
oracle ops in NumPy, and the PCN train / val / test entry points end to end on
synthetic data."""
import math
import os
import sys

import numpy as np
import pytest
import torch
from conftest import ROOT, rand_clouds

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
COMPLETION = os.path.join(ROOT, "completion")
if COMPLETION not in sys.path:
    sys.path.insert(0, COMPLETION)


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


def conv_ref(x, w, b=None):
    """A 1x1 convolution as a batched GEMM (rocBLAS) -- the autograd reference of the tests below.  NOT F.conv1d / F.conv2d:
    MIOpen's implicit-GEMM backward-data kernel of this image (igemm_bwd_gtcx35_nhwc_fp32) reads past its operands on
    some shapes and faults whenever the neighbouring page is unmapped (profiles/NOTES_r6.md section 11)."""
    y = torch.matmul(w.flatten(1), x.flatten(2)).view(x.size(0), w.size(0), *x.shape[2:])
    return y if b is None else y + b.view(1, -1, *([1] * (x.dim() - 2)))


def test_calc_cd_and_calc_emd_formulas(oracle):
    import model_utils as mu
    out, gt = rand_clouds(0, 3, 1024, 3), rand_clouds(1, 3, 2048, 3)
    cd_p, cd_t, f1 = mu.calc_cd(dev(out), dev(gt), calc_f1=True)
    d1, d2, _, _ = oracle.chamfer_forward(gt, out)          # gt is xyz1 (model_utils.py:70)
    np.testing.assert_allclose(cd_p.cpu().numpy(), (np.sqrt(d1).mean(1) + np.sqrt(d2).mean(1)) / 2, rtol=1e-6)
    np.testing.assert_allclose(cd_t.cpu().numpy(), d1.mean(1) + d2.mean(1), rtol=1e-6)
    p1, p2 = (d1 < 1e-4).mean(1), (d2 < 1e-4).mean(1)
    want = np.where(p1 + p2 > 0, 2 * p1 * p2 / np.maximum(p1 + p2, 1e-30), 0)
    np.testing.assert_allclose(f1.cpu().numpy(), want, rtol=1e-5, atol=1e-7)
    out2 = rand_clouds(2, 2, 1024, 3)
    gt2 = rand_clouds(3, 2, 1024, 3)
    e = mu.calc_emd(dev(out2), dev(gt2))                    # eps 0.005, 50 iterations
    od, _ = oracle.emd_forward(out2, gt2, 0.005, 50)
    np.testing.assert_allclose(e.cpu().numpy(), np.sqrt(od).mean(1), rtol=1e-6)


def test_coordinate_knn_goes_through_the_op(oracle):
    """model_utils.knn / knn_point_idx on coordinates use the
    fused knn operator.  Pinned three ways: (1) bit-identical to the oracle's
    knn (the operator's own contract), (2) the chosen neighbours are the true k
    nearest in float64 up to 1e-6 relative on the k-th distance, (3) they agree
    with the reference's matmul + topk formulation wherever consecutive
    neighbour distances differ by more than that formulation's rounding
    (2e-6 absolute for points in the unit cube)."""
    import model_utils as mu
    B, N, M, k = 3, 1536, 512, 16
    pts, ctr = rand_clouds(5, B, N, 3), rand_clouds(6, B, M, 3)
    idx = mu.knn(dev(np.ascontiguousarray(pts.transpose(0, 2, 1))), k)              # (B,N,k), self included
    assert idx.dtype == torch.int64 and idx.shape == (B, N, k)
    np.testing.assert_array_equal(idx.cpu().numpy(), oracle.knn(k, pts).transpose(0, 2, 1))
    assert (idx[:, :, 0].cpu() == torch.arange(N)).all()
    pn = mu.knn_point_idx(k, dev(pts), dev(ctr))                                    # (B,M,k)
    np.testing.assert_array_equal(pn.cpu().numpy(), oracle.knn(k, pts, ctr).transpose(0, 2, 1))
    d2 = ((ctr[:, :, None].astype(np.float64) - pts[:, None].astype(np.float64)) ** 2).sum(-1)
    chosen = np.take_along_axis(d2, pn.cpu().numpy(), -1)
    want = np.sort(d2, -1)[..., :k]
    np.testing.assert_allclose(chosen, want, rtol=1e-6, atol=1e-12)                  # ascending, true k nearest
    ref_idx = mu.knn_point(k, dev(pts), dev(ctr))[1].cpu().numpy()                   # matmul + topk
    differ = ref_idx != pn.cpu().numpy()
    gap = np.abs(np.take_along_axis(d2, ref_idx, -1) - chosen)
    assert (gap[differ] < 2e-6).all() and differ.mean() < 1e-3
    # feature-space searches: library GEMM + one scan of the Gram matrix; the
    # same indices as topk on the materialised matrix wherever values differ
    for C, n, kk in [(24, 700, 16), (48, 1024, 16), (256, 300, 4), (5, 40, 40), (24, 3072, 20), (8, 130, 64),
                     (8, 130, 70), (16, 257, 1)]:
        feat = dev(rand_clouds(7 + C, 2, C, n))
        got = mu.knn(feat, kk)
        assert got.shape == (2, n, kk) and got.dtype == torch.int64
        sq = (feat * feat).sum(dim=1, keepdim=True)
        neg = -sq - (-2 * torch.matmul(feat.transpose(2, 1), feat)) - sq.transpose(2, 1)
        want_v, want_i = neg.topk(k=kk, dim=-1)
        got_v = torch.gather(neg, 2, got)
        assert torch.equal(got_v, want_v)                      # same values in the same (descending) order
        distinct = torch.ones_like(want_v, dtype=torch.bool)
        distinct[..., 1:] &= want_v[..., 1:] != want_v[..., :-1]
        distinct[..., :-1] &= want_v[..., :-1] != want_v[..., 1:]
        assert torch.equal(got[distinct], want_i[distinct])     # same indices where no tie is involved


def test_edge_feature_gathers_match_advanced_indexing():
    """get_edge_features / get_graph_feature gather through the grouping
    operator on the GPU; values and gradients equal the reference's advanced
    indexing formulation (kept for CPU tensors)."""
    import model_utils as mu
    B, C, N, k = 2, 24, 300, 8
    x_cpu = torch.from_numpy(rand_clouds(8, B, C, N)).requires_grad_()
    idx = torch.from_numpy(np.random.default_rng(9).integers(0, N, (B, N, k)))
    x_gpu = x_cpu.detach().to(DEV).requires_grad_()
    want, got = mu.get_edge_features(x_cpu, idx), mu.get_edge_features(x_gpu, idx.to(DEV))
    assert got.shape == (B, C, k, N) and torch.equal(got.cpu(), want)
    w = torch.from_numpy(rand_clouds(10, B, C, k, N))
    (want * w).sum().backward()
    (got * w.to(DEV)).sum().backward()
    assert torch.allclose(x_gpu.grad.cpu(), x_cpu.grad, rtol=1e-5, atol=1e-6)
    # graph feature on coordinates: same neighbours either way for this cloud
    p_cpu = torch.from_numpy(rand_clouds(11, B, 3, 200))
    g_cpu, g_gpu = mu.get_graph_feature(p_cpu, k=6), mu.get_graph_feature(p_cpu.to(DEV), k=6)
    assert g_gpu.shape == (B, 6, 200, 6)
    assert torch.allclose(g_gpu.cpu(), g_cpu, rtol=0, atol=0)


def test_edge_preserve_sampling_composition(oracle):
    import model_utils as mu
    B, C, N, S, k = 2, 16, 768, 384, 10
    feat, pts = rand_clouds(0, B, C, N), rand_clouds(1, B, N, 3)
    net, p_idx, pn_idx, point_output = mu.edge_preserve_sampling(dev(feat), dev(pts), S, k)
    o_idx = oracle.furthest_point_sample(pts, S)
    np.testing.assert_array_equal(p_idx.cpu().numpy(), o_idx)
    o_pts = oracle.gather_points(np.ascontiguousarray(pts.transpose(0, 2, 1)), o_idx).transpose(0, 2, 1)
    np.testing.assert_array_equal(point_output.cpu().numpy(), o_pts)
    pn = pn_idx.cpu().numpy()                                # torch top-k indices: reuse them
    nb = oracle.gather_points(feat, pn.reshape(B, S * k)).reshape(B, C, S, k).max(3)
    ctr = oracle.grouping_operation(feat, o_idx[:, :, None]).reshape(B, C, S)
    np.testing.assert_array_equal(net.cpu().numpy(), np.concatenate([ctr, nb], 1))
    # the torch kNN picked true nearest neighbours
    d2 = ((o_pts[:, :, None] - pts[:, None]) ** 2).sum(-1)
    assert (np.sort(np.take_along_axis(d2, pn.astype(np.int64), -1), -1)[..., -1]
            <= np.sort(d2, -1)[..., k - 1] * (1 + 1e-4) + 1e-7).all()


def test_three_nn_upsampling_and_symmetric_sample(oracle):
    import model_utils as mu
    tgt, src = rand_clouds(0, 2, 768, 3), rand_clouds(1, 2, 384, 3)
    idx, w = mu.three_nn_upsampling(dev(tgt), dev(src))
    od, oi = oracle.three_nn(tgt, src)
    np.testing.assert_array_equal(idx.cpu().numpy(), oi)
    inv = 1.0 / np.maximum(od, 1e-10)
    np.testing.assert_allclose(w.cpu().numpy(), inv / inv.sum(2, keepdims=True), rtol=1e-5)
    sym = mu.symmetric_sample(dev(tgt), 64)
    i = oracle.furthest_point_sample(tgt, 64)
    half = np.take_along_axis(tgt, i[..., None].astype(np.int64), 1)
    np.testing.assert_array_equal(sym.cpu().numpy(), np.concatenate([half, half * [1, 1, -1]], 1).astype(np.float32))


def test_uniform_and_repulsion_losses_run(oracle):
    import model_utils as mu
    pcd = dev(rand_clouds(0, 2, 1024, 3)).requires_grad_()
    loss = mu.get_uniform_loss(pcd)
    assert torch.isfinite(loss) and loss.item() > 0
    loss.backward()
    assert torch.isfinite(pcd.grad).all() and pcd.grad.abs().sum() > 0
    # first percentage, composed from oracle ops: same seeds / balls
    x = pcd.detach().cpu().numpy()
    seeds = oracle.furthest_point_sample(x, int(1024 * 0.05))
    new_xyz = np.take_along_axis(x, seeds[..., None].astype(np.int64), 1)
    idx = oracle.ball_query(0, math.sqrt(0.004), int(1024 * 0.004), x, new_xyz)
    from mvp_benchmark_amd.mm3d_pn2 import ball_query, furthest_point_sample
    np.testing.assert_array_equal(ball_query(0, math.sqrt(0.004), 4, pcd.detach(), dev(new_xyz)).cpu().numpy(), idx)
    rep = mu.get_repulsion_loss(dev(rand_clouds(1, 2, 512, 3)))
    assert torch.isfinite(rep)


def test_uniform_and_repulsion_loss_values(oracle):
    """get_uniform_loss / get_repulsion_loss (model_utils.py:181-227) against a
    composition of the oracle's FPS / ball_query / group / knn.

    The uniform loss takes the nearest-neighbour spacing inside a ball from
    knn_point's float32 matmul formulation -|x|^2 + 2 x.y - |y|^2.  ball_query pads
    short balls with their first hit, so 2/3 of the slots are exact duplicates whose
    "distance" is that formulation's cancellation noise (|.| ~ 1e-7 instead of 0,
    under sqrt(|. + 1e-8|)): the reference's value (0.031 here) is not the float64
    value of the same formula (0.022).  The check therefore replays the reference's
    float32 formulation (CPU torch) on the oracle-composed balls; the tolerance
    covers the rounding-order difference between the CPU and GPU matmul on those
    noise terms."""
    import model_utils as mu
    x = rand_clouds(5, 2, 1024, 3)
    B, N, _ = x.shape
    got = float(mu.get_uniform_loss(dev(x)))
    seeds = oracle.furthest_point_sample(x, int(N * 0.05))
    new_xyz = np.take_along_axis(x, seeds[..., None].astype(np.int64), 1)
    xt = np.ascontiguousarray(x.transpose(0, 2, 1))
    want, exact = 0.0, 0.0
    for p in [0.004, 0.006, 0.008, 0.010, 0.012]:
        nsample, r = int(N * p), math.sqrt(p)
        expect_len = math.sqrt(math.pi * p / nsample)
        idx = oracle.ball_query(0, r, nsample, x, new_xyz)
        g = torch.tensor(oracle.grouping_operation(xt, idx).transpose(0, 2, 3, 1).reshape(-1, nsample, 3))
        inner = -2 * torch.matmul(g, g.transpose(2, 1))                  # knn_point(2, g, g), model_utils.py:250-259
        sq = (g ** 2).sum(2)
        var = (-sq.unsqueeze(2) - inner - sq.unsqueeze(1)).topk(2, dim=-1)[0]
        dis = torch.sqrt(torch.abs(-var[:, :, 1:] + 1e-8)).mean(-1)
        want += float(((dis - expect_len) ** 2 / (expect_len + 1e-8)).mean()) * (p * 100) ** 2
        d2 = ((g.double()[:, :, None] - g.double()[:, None]) ** 2).sum(-1).sort(-1)[0][:, :, 1]
        dis64 = torch.sqrt(torch.abs(d2 + 1e-8)).mean(-1)
        exact += float(((dis64 - expect_len) ** 2 / (expect_len + 1e-8)).mean()) * (p * 100) ** 2
    want, exact = want / 5, exact / 5
    assert got == pytest.approx(want, rel=2e-2)
    assert abs(want - exact) > 0.2 * exact            # (documents the float32 effect described above)

    y = rand_clouds(6, 2, 512, 3)
    got = float(mu.get_repulsion_loss(dev(y)))
    idx = oracle.knn(20, y, y).transpose(0, 2, 1)                     # (B, N, 20), self first
    nb = np.take_along_axis(y[:, None].astype(np.float64), idx[..., None].astype(np.int64), 2) - y[:, :, None]
    d2 = np.sort((nb ** 2).sum(-1), axis=-1)[:, :, 1:5]              # 4 nearest besides the point itself
    d2 = np.maximum(d2, 1e-12)
    want = np.mean(0.07 - np.sqrt(d2) * np.exp(-d2 / 0.03 ** 2))
    assert got == pytest.approx(want, rel=1e-4)


def _pcn_args(tmp, **kw):
    import train
    args = train.load_config(os.path.join(COMPLETION, "cfgs", "pcn.yaml"))
    args.update(batch_size=4, nepoch=2, work_dir=str(tmp), synthetic=True, synthetic_train_shapes=1,
                synthetic_val_shapes=1, manual_seed=7, step_interval_to_print=1000, lr=1e-3)
    args.update(kw)
    return args


def test_pcn_train_val_test_end_to_end(tmp_path):
    """completion/train.py + test.py on synthetic MVP-shaped data: losses are
    finite and go down, val returns the reference's metric names, the
    checkpoint round-trips into test.py which writes the submission array."""
    import test as test_entry
    import train
    args = _pcn_args(tmp_path, eval_emd=True)
    log_dir = str(tmp_path / "run")
    os.makedirs(log_dir)
    metrics = train.train(args, log_dir, "pcn_test")
    assert set(metrics) == {'cd_p', 'cd_t', 'emd', 'f1'}
    assert all(math.isfinite(v) for v in metrics.values())
    assert 0 < metrics['cd_t'] < 1 and 0 < metrics['emd'] < 1
    assert os.path.exists(os.path.join(log_dir, "network.pth"))
    assert os.path.exists(os.path.join(log_dir, "best_cd_t_network.pth"))
    args2 = _pcn_args(tmp_path, load_model=os.path.join(log_dir, "network.pth"))
    res = test_entry.test(args2, log_dir)
    assert res.shape == (26, 2048, 3) and np.isfinite(res).all()
    assert os.path.exists(os.path.join(log_dir, "results.npy")) or os.path.exists(os.path.join(log_dir, "results.h5"))


def test_pcn_training_reduces_chamfer_loss():
    import train
    from models import pcn
    args = _pcn_args("/tmp")
    torch.manual_seed(0)
    net = pcn.Model(args).to(DEV)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(0)
    gt = torch.rand(4, 2048, 3, generator=g).to(DEV) * 0.5
    x = gt[:, :2048].transpose(2, 1).contiguous()
    losses = []
    for _ in range(30):
        opt.zero_grad()
        _, _, loss = net(x, gt, alpha=1.0)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(math.isfinite(v) for v in losses)
    assert losses[-1] < 0.5 * losses[0]


def test_graphed_training_step_matches_eager():
    """train.GraphedStep (cfg key `hip_graph`): the step replayed from one HIP graph trains the
    same network as the eager step -- PCN (no random numbers in its forward), same initial
    weights, same batches, alpha and learning rate changed between replays: losses and weights
    agree after six steps."""
    import train
    from models import pcn
    args = _pcn_args("/tmp")
    g = torch.Generator().manual_seed(3)
    batches = [(torch.rand(4, 2048, 3, generator=g).to(DEV) * 0.5) for _ in range(3)]

    def run(graphed):
        torch.manual_seed(0)
        net = pcn.Model(args).to(DEV).train()
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True)     # same update arithmetic both ways
        step = train.GraphedStep(net, opt, torch.device(DEV)) if graphed else None
        losses = []
        # (the captured variant's three warm-up steps are rolled back -- parameters, optimizer state, random
        # streams: ADVICE r3 -- so both variants train on every batch exactly once)
        for it in range(6):
            gt = batches[it % 3]
            x = gt.transpose(2, 1).contiguous()
            alpha = 0.5 if it < 3 else 1.0
            lr = 1e-3 if it < 4 else 5e-4
            if graphed:
                step.set_lr(lr)
                _, total = step(x, gt, alpha)
                losses.append(float(total))
            else:
                for group in opt.param_groups:
                    group['lr'] = torch.tensor(lr, device=DEV)
                opt.zero_grad()
                _, _, loss = net(x, gt, alpha=alpha)
                loss.mean().backward()
                opt.step()
                losses.append(float(loss.mean()))
        return losses, [p.detach().clone() for p in net.parameters()]

    le, pe = run(False)
    lg, pg = run(True)
    # (Adam's first steps move every weight by ~lr whatever the gradient's size, so the float-atomic
    # noise of the Chamfer gradient is amplified: per-mille agreement is what identical steps give;
    # a step that ignored the new alpha / learning rate would be off by tens of per cent)
    np.testing.assert_allclose(lg, le, rtol=2e-2)
    assert abs(lg[3] - lg[2]) > 5 * abs(lg[3] - le[3])        # the alpha change at step 3 is visible in both
    # weights: on average far closer than the ~9e-3 nine Adam steps of lr 1e-3 can move one
    # (single weights with a near-zero, noise-dominated gradient may go opposite ways)
    num = sum(float((a - b).abs().sum()) for a, b in zip(pg, pe))
    assert num / sum(a.numel() for a in pg) < 5e-4


def test_pcn_eval_config2_shapes():
    """BASELINE config 2: PCN eval 2048 -> 16384 points, batch 32: CD + F1 on
    the network output, EMD (eval settings) between two spread 16384-point
    clouds of the same batch.  (A random-init PCN emits one tight cluster; an
    auction between a cluster and a spread cloud needs orders of magnitude
    more bids -- that stress case is test_emd_clustered_prediction_stress.)"""
    import model_utils as mu
    import train
    from models import pcn
    args = train.load_config(os.path.join(COMPLETION, "cfgs", "pcn_eval16k.yaml"))
    args.eval_emd = False
    torch.manual_seed(1)
    net = pcn.Model(args).to(DEV).eval()
    g = torch.Generator().manual_seed(0)
    partial = torch.rand(32, 3, 2048, generator=g).to(DEV)
    gt = torch.rand(32, 16384, 3, generator=g).to(DEV)
    with torch.no_grad():
        r = net(partial, gt, prefix="val")
        e = mu.calc_emd(torch.rand(32, 16384, 3, generator=g).to(DEV), gt, eps=0.004, iterations=3000)
    assert r['out2'].shape == (32, 16384, 3)
    for k in ('cd_p', 'cd_t', 'f1'):
        assert r[k].shape == (32,) and torch.isfinite(r[k]).all(), k
    assert e.shape == (32,) and torch.isfinite(e).all()


def test_emd_clustered_prediction_stress(oracle):
    """Worst case for the pruned search: every person sits in one tight
    cluster (what an untrained PCN emits), the objects are spread.  Prices
    rise everywhere, the search cube covers the grid and the kernel falls back
    to its linear scan; results stay bit-identical to the oracle."""
    from mvp_benchmark_amd.metrics import emd
    x1 = (0.5 + 0.01 * rand_clouds(0, 2, 2048, 3)).astype(np.float32)
    x2 = rand_clouds(1, 2, 2048, 3)
    dist, ass = emd()(dev(x1), dev(x2), 0.004, 300)
    od, oa = oracle.emd_forward(x1, x2, 0.004, 300)
    np.testing.assert_array_equal(ass.cpu().numpy(), oa)
    np.testing.assert_array_equal(dist.cpu().numpy(), od)


@pytest.mark.parametrize("name", ["ecg", "vrcnet"])
def test_ecg_vrcnet_train_val_test_steps(name):
    """BASELINE config 3 family: one optimisation step + val + test forward of
    ECG / VRCNet on MVP-shaped synthetic clouds; every op of the layer is
    exercised forward and backward (FPS, gather, group, three_nn/interpolate,
    ball_query, CD, EMD)."""
    import importlib
    import train
    args = train.load_config(os.path.join(COMPLETION, "cfgs", name + ".yaml"))
    args.update(eval_emd=True, load_model=None)
    torch.manual_seed(3)
    net = importlib.import_module("models." + name).Model(args).to(DEV)
    g = torch.Generator().manual_seed(0)
    gt = torch.rand(2, 2048, 3, generator=g).to(DEV)
    partial = gt[:, torch.randperm(2048, generator=g)[:2048]].transpose(2, 1).contiguous()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    net.train()
    losses = []
    for _ in range(2):
        opt.zero_grad()
        fine, loss_fine, total = net(partial, gt, alpha=0.5)
        total.backward()
        opt.step()
        losses.append(total.item())
    assert all(math.isfinite(v) for v in losses)
    expect_b = 4 if name == "vrcnet" else 2          # VRCNet doubles the batch in training
    assert fine.shape == (expect_b, 2048, 3) and loss_fine.shape == (expect_b,)
    grads = [p.grad for p in net.parameters() if p.grad is not None]
    assert grads and all(torch.isfinite(gr).all() for gr in grads)
    net.eval()
    with torch.no_grad():
        r = net(partial, gt, prefix="val")
        t = net(partial, prefix="test")
    assert t['result'].shape == (2, 2048, 3)
    for k in ('cd_p', 'cd_t', 'f1', 'emd'):
        assert r[k].shape == (2,) and torch.isfinite(r[k]).all(), k


def test_vrcnet_full_fps_of_gt_changes_nothing(monkeypatch):
    """VRCNet's training path feeds its PointNet encoder with gt re-ordered by an FPS of ALL its points
    (reference completion/models/vrcnet.py:451).  Shown here: (i) that FPS returns a permutation; (ii) the
    encoder's output for gt, for gt in FPS order and for a random permutation agree to float32 rounding
    (per-point maps and max-pools cannot see the order) -- bit for bit on the op layer's convolution kernels;
    (iii) a training forward without that FPS (the default since round 4) and one with the reference's sequence
    (op_config skip_full_fps_of_gt = False) return the same loss and CD to 1e-4 / 1e-3 and the same fine clouds as point sets
    (same seed for the latent samples; the decoder re-orders its own points by the order of its input)."""
    import importlib
    import train
    from mvp_benchmark_amd.mm3d_pn2 import furthest_point_sample, gather_points
    args = train.load_config(os.path.join(COMPLETION, "cfgs", "vrcnet.yaml"))
    args.update(load_model=None)
    torch.manual_seed(5)
    net = importlib.import_module("models.vrcnet").Model(args).to(DEV).train()
    g = torch.Generator().manual_seed(1)
    gt = torch.rand(4, 2048, 3, generator=g).to(DEV)
    partial = gt[:, torch.randperm(2048, generator=g)].transpose(2, 1).contiguous()
    with torch.no_grad():
        plain = net.encoder(gt.transpose(1, 2).contiguous())
        order = furthest_point_sample(gt, 2048)
        assert sorted(order[0].tolist()) == list(range(2048))                       # a permutation
        by_fps = net.encoder(gather_points(gt.transpose(1, 2).contiguous(), order))
        shuffled = net.encoder(gt[:, torch.randperm(2048, generator=g)].transpose(1, 2).contiguous())
    assert torch.equal(plain, by_fps) and torch.equal(plain, shuffled)
    outs = []
    for skip in (False, True):
        import op_config
        monkeypatch.setattr(op_config.OPS, "skip_full_fps_of_gt", skip)
        torch.manual_seed(11)
        with torch.no_grad():
            outs.append(net(partial, gt, alpha=0.5))
    # (the network amplifies the encoder's last-bit differences, and an index decision -- the decoder's own FPS, a
    # top-k -- may flip for single points: compare in the mean)
    (fine_a, cd_a, loss_a), (fine_b, cd_b, loss_b) = outs
    assert torch.allclose(cd_a, cd_b, rtol=1e-3) and abs(float(loss_a) - float(loss_b)) <= 1e-4 * abs(float(loss_a))
    from model_utils import calc_cd
    assert float(calc_cd(fine_a, fine_b)[0].max()) <= 1e-3 * float(cd_a.min())      # the same clouds, as point sets


@pytest.mark.parametrize("B,share,Cw,k,N", [(2, 8, 2, 16, 3072), (3, 8, 16, 10, 384), (1, 4, 3, 5, 77), (2, 1, 5, 3, 300),
                                            (2, 16, 1, 20, 257), (1, 2, 7, 1, 64)])
def test_share_weighted_sum_matches_torch(B, share, Cw, k, N):
    """The fused neighbourhood aggregation of SA_module against the plain
    PyTorch fp32 formulation of vrcnet.py:52-55 (repeat, product, sum over k):
    forward and both gradients, 1e-5 relative (the summation order over k differs
    from torch's reduction)."""
    import model_utils as mu
    from mvp_benchmark_amd.mm3d_pn2.functional import share_weighted_sum
    g = torch.Generator().manual_seed(B * 1000 + N)
    w = torch.randn(B, Cw, k, N, generator=g).to(DEV).requires_grad_()
    v = torch.randn(B, share * Cw, k, N, generator=g).to(DEV).requires_grad_()
    go = torch.randn(B, share * Cw, N, generator=g).to(DEV)
    out = share_weighted_sum(w, v)
    ref = (w.repeat(1, share, 1, 1) * v).sum(dim=2)
    assert out.shape == ref.shape
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-5)
    gw, gv = torch.autograd.grad(out, (w, v), go)
    rw, rv = torch.autograd.grad(ref, (w, v), go)
    assert torch.allclose(gv, rv, rtol=1e-5, atol=1e-6)
    assert torch.allclose(gw, rw, rtol=1e-5, atol=1e-5)
    # the models' entry point: op layer on the GPU, the broadcast formulation elsewhere
    got = mu.aggregate_shared(w, v, share)
    cpu = mu.aggregate_shared(w.detach().cpu().double(), v.detach().cpu().double(), share)
    assert torch.allclose(got.detach().cpu().double(), cpu, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("shape,cout,bias", [((32, 48, 16, 1024), 24, True), ((4, 24, 16, 3072), 24, False),
                                             ((8, 256, 1, 768), 16, True), ((8, 64, 3072), 4, True),
                                             ((3, 16, 1, 300), 64, True), ((2, 512, 1, 384), 32, False),
                                             ((2, 130, 5, 44), 33, True), ((1, 1, 4), 1, True), ((2, 7, 1028), 5, True),
                                             ((3, 64, 16, 516), 64, True), ((2, 33, 2052), 49, False), ((5, 17, 3, 8), 9, True)])
def test_pointwise_conv_weight_gradient(shape, cout, bias):
    """Per-point / per-edge linear maps: weight and bias gradients from
    mvp_pointwise_wgrad against PyTorch's fp32 convolution_backward at 1e-4 of
    the gradient's scale (different summation order over ~1e5..1e6 terms); output
    and input gradient identical to the library convolution where it IS the
    library convolution, within fp32 summation order where the MFMA GEMM runs
    (>= 32 input and output channels)."""
    import torch.nn.functional as F
    from mvp_benchmark_amd import _lib
    from mvp_benchmark_amd.pointwise import MAX_CIN, MAX_COUT, MFMA_MIN_CH, pointwise_conv, _PointwiseConv
    g = torch.Generator().manual_seed(shape[1] * 131 + cout)
    x = torch.randn(*shape, generator=g).to(DEV).requires_grad_()
    w = torch.randn(cout, shape[1], *([1] * (len(shape) - 2)), generator=g).to(DEV).requires_grad_()
    b = torch.randn(cout, generator=g).to(DEV).requires_grad_() if bias else None
    conv = F.conv1d if len(shape) == 3 else F.conv2d
    with torch.no_grad():
        ref = conv(x, w, b)                       # (forward only: the library's backward-data kernel is not trusted, conv_ref)
    ref_g = conv_ref(x, w, b)
    go = torch.randn_like(ref)
    params = (x, w) + ((b,) if bias else ())
    want = torch.autograd.grad(ref_g, params, go)
    close = lambda a_, b_: torch.allclose(a_, b_, rtol=1e-4, atol=1e-4 * float(b_.abs().max()))
    # the kernel itself, every shape (the models only route layers with <= MAX_CIN input channels to it)
    B, cin, length = shape[0], shape[1], x[0, 0].numel()
    nbytes = _lib.pointwise_wgrad_scratch_bytes(B, cin, cout, length)
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    gw, gb = torch.full_like(w, float("nan")), torch.full((cout,), float("nan"), device=DEV)
    _lib.call("mvp_pointwise_wgrad", DEV, B, cin, cout, length, x.detach(), go, gw, gb if bias else None, ws, nbytes)
    assert close(gw, want[1]) and (not bias or close(gb, want[2]))
    gw2 = torch.empty_like(w)
    _lib.call("mvp_pointwise_wgrad", DEV, B, cin, cout, length, x.detach(), go, gw2, None, ws, nbytes)
    assert torch.equal(gw, gw2)                                          # fixed summation order: reproducible
    # the autograd route
    import mvp_benchmark_amd.pointwise as pw
    assert pw.MFMA_TRAIN                                                 # the default since round 4
    y = pointwise_conv(x, w, b)
    mfma = cin >= MFMA_MIN_CH and cout >= MFMA_MIN_CH and length % 4 == 0
    small = cin <= MAX_CIN and cout <= MAX_COUT and length % 4 == 0
    # (round 6: EVERY float32 CUDA layer under autograd goes through _PointwiseConv -- also the shapes whose forward is the
    # library's: its backward never calls MIOpen's backward-data kernels, one of which reads out of bounds on odd shapes)
    assert isinstance(y.grad_fn, _PointwiseConv._backward_cls)
    pw.MFMA_TRAIN = False
    try:
        y_lib = pointwise_conv(x, w, b)                                  # switched off: the library's forward unless small
    finally:
        pw.MFMA_TRAIN = True
    assert isinstance(y_lib.grad_fn, _PointwiseConv._backward_cls) == (small or not mfma)
    pw.USE_MFMA = False
    try:
        y_plain = pointwise_conv(x, w, b)                                # nothing of ours: plain autograd on the library
    finally:
        pw.USE_MFMA = True
    assert isinstance(y_plain.grad_fn, _PointwiseConv._backward_cls) == small      # (the few-channel kernels are not MFMA kernels)
    got = torch.autograd.grad(y, params, go)
    if mfma:
        assert close(y, ref) and close(got[0], want[0])
    elif small:
        # few channels: the data gradient is mvp_pointwise_dgrad (fp32, another summation order over <= 64 terms)
        assert torch.equal(y, ref) and close(got[0], want[0])
    else:
        assert torch.equal(y, ref) and close(got[0], want[0])          # (the data gradient is a batched GEMM, not the library's convolution)
    if cin <= 64 and cout <= 64 and length % 4 == 0:
        gx = torch.full_like(x, float("nan"))
        _lib.call("mvp_pointwise_dgrad", DEV, B, cin, cout, length, w.detach(), go, gx)
        assert close(gx, want[0])
    for a_, b_ in zip(got[1:], want[1:]):
        assert close(a_, b_)
    # a length that is not a multiple of 4 is outside every kernel's cover: the library's forward, GEMM-formulated gradients
    if shape[-1] > 1:
        x3 = x.detach()[..., :-1].contiguous().requires_grad_()
        y3 = pointwise_conv(x3, w, b)
        assert isinstance(y3.grad_fn, _PointwiseConv._backward_cls)
        r3 = conv_ref(x3, w, b)
        go3 = torch.randn_like(r3)
        for a_, b_ in zip(torch.autograd.grad(y3, (x3,) + tuple(params[1:]), go3), torch.autograd.grad(r3, (x3,) + tuple(params[1:]), go3)):
            assert close(a_, b_)


def test_ef_expansion_on_the_op_layer():
    """EF_expansion (ECG / VRCNet up-sampling heads, scale >= 2) on the GPU:
    knn through the Gram top-k, gathers through the grouping operator; output
    and gradients against the reference's edge-tensor formulation in fp32."""
    import torch.nn.functional as F
    import model_utils as mu
    torch.manual_seed(2)
    B, C, N, k, out, step = 2, 64, 512, 4, 16, 2
    ef = mu.EF_expansion(C, output_size=out, step_ratio=step, k=k).to(DEV)
    x = torch.randn(B, C, N, device=DEV, requires_grad=True)
    got = ef(x)
    edge_in = mu.get_graph_feature(x, k, minus_center=False).permute(0, 1, 3, 2).contiguous()
    edge = F.relu(torch.cat((ef.conv1(edge_in), edge_in), 1))
    edge = F.relu(ef.conv2(edge))
    edge = edge.permute(0, 2, 3, 1).contiguous().view(B, k, N * step, out).permute(0, 3, 1, 2)
    ref = ef.conv3(edge).max(dim=2)[0]
    assert got.shape == ref.shape == (B, out, N * step)
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-4)
    params = [x] + list(ef.parameters())
    for a, b in zip(torch.autograd.grad(got.square().sum(), params), torch.autograd.grad(ref.square().sum(), params)):
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-3 * float(b.abs().max()))


@pytest.mark.parametrize("B,cin,cout,L", [(2, 128, 256, 2048), (3, 256, 64, 384), (2, 515, 128, 1536), (1, 1090, 256, 2048),
                                          (2, 64, 64, 3072), (2, 32, 48, 260), (2, 200, 300, 124), (1, 40, 33, 8),
                                          (2, 3, 128, 2048), (2, 68, 2, 3072), (2, 544, 16, 384), (2, 8, 128, 768), (3, 2, 1, 64)])
def test_pointwise_mfma_matches_torch(B, cin, cout, L):
    """mvp_pointwise_mfma (float32 MFMA, k-ordered fmaf chain) against the plain
    PyTorch fp32 convolution: y = W x (+ bias, ReLU, residual), and the data
    gradient W^T g through the same kernel (w_kmajor).  Tolerance: fp32 summation
    order only -- 2e-6 * sqrt(K) relative to the operands' scale (unit normals)."""
    import torch.nn.functional as F
    from mvp_benchmark_amd.pointwise import mfma_linear
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = torch.randn(B, cin, L, generator=g).to(DEV)
    w = torch.randn(cout, cin, generator=g).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    res = torch.randn(B, cout, L, generator=g).to(DEV)
    tol = 4e-6 * math.sqrt(cin) * 4
    ref = F.conv1d(x.double(), w.double().unsqueeze(2), b.double())
    got = mfma_linear(x, w, b)
    assert (got.double() - ref).abs().max().item() < tol
    got = mfma_linear(x, w, b, relu=True, residual=res)
    assert (got.double() - (torch.relu(ref) + res.double())).abs().max().item() < tol
    got = mfma_linear(x, w)                                                     # no bias
    assert (got.double() - F.conv1d(x.double(), w.double().unsqueeze(2))).abs().max().item() < tol
    if cin % 4 == 0:                                                            # data gradient: W^T gy
        gy = torch.randn(B, cout, L, generator=g).to(DEV)
        gx = mfma_linear(gy, w, w_kmajor=True)
        want = torch.einsum("oc,bol->bcl", w.double(), gy.double())
        assert (gx.double() - want).abs().max().item() < 4e-6 * math.sqrt(cout) * 4


@pytest.mark.parametrize("B,cin,cout,L,bias", [(4, 128, 256, 768, True), (64, 64, 128, 3072, False), (3, 515, 130, 388, True),
                                               (2, 1090, 256, 2048, True), (5, 40, 33, 16, True), (1, 256, 64, 20, False),
                                               (4, 3, 128, 2048, True), (4, 68, 2, 3072, True), (4, 272, 8, 768, True),
                                               (64, 512, 1024, 512, True), (2, 1, 1, 4, True)])
def test_pointwise_wgrad_mfma_matches_torch(B, cin, cout, L, bias):
    """mvp_pointwise_wgrad_mfma (weight + bias gradient as one MFMA GEMM over all positions, partial
    tiles summed in a fixed order) against float64; with the ReLU mask; bit-reproducible."""
    from mvp_benchmark_amd.pointwise import mfma_wgrad
    g = torch.Generator().manual_seed(cin + cout + L)
    x = torch.randn(B, cin, L, generator=g).to(DEV)
    gy = torch.randn(B, cout, L, generator=g).to(DEV)
    y = torch.randn(B, cout, L, generator=g).to(DEV)
    gw, gb = mfma_wgrad(x, gy, cout, cin, bias)
    want = torch.einsum("bol,bil->oi", gy.double(), x.double())
    tol = 3e-6 * math.sqrt(B * L) * 4
    assert (gw.double() - want).abs().max().item() < tol
    if bias:
        assert (gb.double() - gy.double().sum((0, 2))).abs().max().item() < tol
    gw2, gb2 = mfma_wgrad(x, gy, cout, cin, bias, gymask=y)
    gm = gy.double() * (y > 0)
    assert (gw2.double() - torch.einsum("bol,bil->oi", gm, x.double())).abs().max().item() < tol
    if bias:
        assert (gb2.double() - gm.sum((0, 2))).abs().max().item() < tol
    gw3, _ = mfma_wgrad(x, gy, cout, cin, bias)
    assert torch.equal(gw, gw3)


@pytest.mark.parametrize("B,cin,cout,L", [(2, 128, 256, 2048), (3, 16, 64, 3072), (2, 68, 2, 3072), (2, 512, 512, 384),
                                          (2, 515, 130, 388), (1, 40, 33, 8)])
def test_pointwise_mfma_ex_prologue_and_epilogues_equal_the_separate_passes(B, cin, cout, L):
    """mvp_pointwise_mfma_ex (ABI 18): every fused step is the exact float operation of the separate elementwise pass, so
    each flag combination must equal -- BIT FOR BIT -- the plain GEMM (mvp_pointwise_mfma's arithmetic) between the
    corresponding torch passes: ReLU of x on load, a bias per cloud, + residual then ReLU, the residual as a mask (the data
    gradient of conv(relu(x))), two outputs."""
    from mvp_benchmark_amd.pointwise import mfma_linear
    g = torch.Generator().manual_seed(cin * 11 + cout)
    x = torch.randn(B, cin, L, generator=g).to(DEV)
    w = torch.randn(cout, cin, generator=g).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    cb = torch.randn(B, cout, generator=g).to(DEV)
    res = torch.randn(B, cout, L, generator=g).to(DEV)
    plain = mfma_linear(x, w)
    assert torch.equal(mfma_linear(x, w, x_relu=True), mfma_linear(torch.relu(x), w))
    assert torch.equal(mfma_linear(x, w, b, x_relu=True, relu=True), torch.relu(mfma_linear(torch.relu(x), w, b)))
    assert torch.equal(mfma_linear(x, w, cb, bias_per_cloud=True), plain + cb.unsqueeze(2))
    assert torch.equal(mfma_linear(x, w, cb, bias_per_cloud=True, relu=True), torch.relu(plain + cb.unsqueeze(2)))
    assert torch.equal(mfma_linear(x, w, b, residual=res, relu_after=True), torch.relu((plain + b.view(1, -1, 1)) + res))
    assert torch.equal(mfma_linear(x, w, b, residual=res, relu=True, relu_after=True),
                       torch.relu(torch.relu(plain + b.view(1, -1, 1)) + res))
    assert torch.equal(mfma_linear(x, w, residual=res, res_is_mask=True), torch.where(res > 0, plain, torch.zeros_like(plain)))
    assert torch.equal(mfma_linear(x, w, relu_after=True), torch.relu(plain))
    if cin % 4 == 0:            # the data gradient of conv(relu(.)) masked by the layer's input, with ReLU' of the output on load
        gy = torch.randn(B, cout, L, generator=g).to(DEV)
        y = torch.randn(B, cout, L, generator=g).to(DEV)
        got = mfma_linear(gy, w, w_kmajor=True, xmask=y, residual=x, res_is_mask=True)
        want = mfma_linear(gy * (y > 0), w, w_kmajor=True) * (x > 0)
        assert torch.equal(got, want + 0.0)          # (+ 0.0: the product's -0 where a masked-out value was negative)
    for split in (32, 64, 96):
        if split < cout:
            y1, y2 = mfma_linear(x, w, b, m_split=split)
            full = mfma_linear(x, w, b)
            assert y1.is_contiguous() and y2.is_contiguous() and y1.shape == (B, split, L) and y2.shape == (B, cout - split, L)
            assert torch.equal(y1, full[:, :split]) and torch.equal(y2, full[:, split:])


def test_pointwise_mfma_ex_rejects_what_it_does_not_cover():
    from mvp_benchmark_amd import _lib
    x = torch.randn(2, 32, 64, device=DEV); w = torch.randn(64, 32, device=DEV); y = torch.empty(2, 64, 64, device=DEV)
    def rc(flags=0, residual=None, m_split=0, y2=None, bias=None, per_cloud=0):
        try:
            _lib.call("mvp_pointwise_mfma_ex", x.device, 2, 32, 64, 64, x, None, w, 0, 0, bias, per_cloud, residual, flags, 1, y, m_split, y2)
            return 0
        except _lib.MvpOpsError as e:
            return str(e)
    assert rc() == 0
    assert rc(flags=16) != 0                                   # unknown flag
    assert rc(flags=4) != 0                                    # a mask without the tensor
    assert rc(m_split=32) != 0                                 # two outputs without the second tensor
    assert rc(m_split=16, y2=y) != 0                           # not a multiple of 32
    assert rc(per_cloud=1) != 0                                # a bias per cloud without the bias


@pytest.mark.parametrize("B,cin,cout,L", [(8, 128, 128, 3072), (8, 16, 64, 3072), (64, 68, 2, 3072), (64, 512, 512, 384), (16, 256, 128, 1536)])
def test_pointwise_conv_fused_matches_the_composed_ops_under_autograd(B, cin, cout, L):
    """pointwise_conv_fused / pointwise_conv_dual (one GEMM with the activations, residual and per-cloud vector inside)
    against the same function composed of pointwise_conv and torch's elementwise passes: outputs bit for bit, every gradient
    at summation-order tolerance (the composed route may use other kernels for a small layer's backward pass)."""
    from mvp_benchmark_amd import pointwise as pw
    g = torch.Generator().manual_seed(cin + 3 * cout)
    def T(*shape, grad=True):
        return torch.randn(*shape, generator=g).to(DEV).requires_grad_(grad)
    x, w, b, res, cb = T(B, cin, 1, L), T(cout, cin, 1, 1), T(cout), T(B, cout, 1, L), T(B, cout)
    go = torch.randn(B, cout, 1, L, generator=g).to(DEV)
    assert pw._fused_routes(x, w, True) == (cin % 4 == 0)
    cases = [dict(relu_in=True), dict(relu_in=True, relu=True), dict(relu_in=True, residual=res, relu_after=True),
             dict(residual=res, relu_after=True), dict(residual=res), dict(relu=True, cloud_bias=cb), dict(cloud_bias=cb),
             dict(relu_in=True, relu=True, bias=None)]
    for kw in cases:
        bias = kw.pop("bias", b)
        inputs = [t for t in (x, w, bias, kw.get("residual"), kw.get("cloud_bias")) if t is not None]
        y = pw.pointwise_conv_fused(x, w, bias, **kw)
        grads = torch.autograd.grad(y, inputs, go)
        a = torch.relu(x) if kw.get("relu_in") else x
        if kw.get("cloud_bias") is not None:            # (the per-cloud vector and the bias are summed first: one addend per output)
            h = pw.pointwise_conv(a, w, None) + (cb + bias if bias is not None else cb).view(B, cout, 1, 1)
        else:
            h = pw.pointwise_conv(a, w, bias)
        h = torch.relu(h) if kw.get("relu") else h
        h = h + res if kw.get("residual") is not None else h
        h = torch.relu(h) if kw.get("relu_after") else h
        want = torch.autograd.grad(h, inputs, go)
        assert torch.equal(y, h), kw
        for got_g, want_g, t in zip(grads, want, inputs):
            scale = want_g.abs().max().item() + 1e-6
            assert (got_g - want_g).abs().max().item() < 2e-5 * scale * math.sqrt(max(cin, cout, L)), (kw, tuple(t.shape))
    if cout % 32 == 0 and cin % 4 == 0:
        w2 = T(cout // 2 if cout > 32 else cout, cin, 1, 1)
        y1, y2 = pw.pointwise_conv_dual(x, w, w2)
        assert y1.is_contiguous() and y2.is_contiguous()
        r1, r2 = pw.pointwise_conv(x, w), pw.pointwise_conv(x, w2)
        assert torch.equal(y1, r1) and torch.equal(y2, r2)
        go2 = torch.randn_like(y2)
        got = torch.autograd.grad([y1, y2], (x, w, w2), [go, go2])
        want = torch.autograd.grad([r1, r2], (x, w, w2), [go, go2])
        for a_, r_ in zip(got, want):
            assert (a_ - r_).abs().max().item() < 2e-5 * (r_.abs().max().item() + 1e-6) * math.sqrt(max(cin, cout, L))


def test_relational_unit_with_fused_activations_equals_the_op_by_op_route():
    """SKN_Res_unit at two of the shipped levels with op_config.fused_activations on / off: same output bits, gradients at
    summation-order tolerance."""
    import op_config
    from models.relational import SKN_Res_unit
    torch.manual_seed(5)
    for cin, c, n in ((4, 64, 3072), (128, 128, 1536)):
        unit = SKN_Res_unit(cin, c, k=[16], layers=1).to(DEV)
        x0 = torch.randn(16, cin, 1, n, device=DEV)
        idx = [torch.randint(0, n, (16, n, 16), device=DEV, dtype=torch.int32)]
        runs = {}
        for on in (False, True):
            old = op_config.configure(fused_activations=on)
            try:
                unit.zero_grad()
                x = x0.clone().requires_grad_()
                out = unit(x, idx, relu_out=True)
                out.square().sum().backward()
                runs[on] = (out.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in unit.named_parameters()})
            finally:
                op_config.configure(**old)
        assert torch.equal(runs[True][0], runs[False][0])
        tol = lambda r: 1e-4 * (r.abs().max().item() + 1e-6)
        assert (runs[True][1] - runs[False][1]).abs().max().item() < tol(runs[False][1])
        for k, gr in runs[False][2].items():
            assert (runs[True][2][k] - gr).abs().max().item() <= tol(gr), k


@pytest.mark.parametrize("group", [2, 4, 16, 32])
def test_pointwise_mfma_group_max(group):
    """The set-abstraction epilogue: conv -> ReLU -> max over the `group`
    neighbours of a point, (B, C, P, S) never written."""
    import torch.nn.functional as F
    from mvp_benchmark_amd.pointwise import mfma_linear
    g = torch.Generator().manual_seed(group)
    B, cin, cout, P = 2, 96, 160, 200
    x = torch.randn(B, cin, P, group, generator=g).to(DEV)
    w = torch.randn(cout, cin, generator=g).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    got = mfma_linear(x, w, b, relu=True, group=group)
    ref = torch.relu(F.conv2d(x.double(), w.double().view(cout, cin, 1, 1), b.double())).max(dim=3)[0]
    assert got.shape == (B, cout, P)
    assert (got.double() - ref).abs().max().item() < 2e-4


def test_pointwise_conv_autograd_through_mfma():
    """PointwiseConv1d / pointwise_conv(relu=True): outputs and all three
    gradients against nn.Conv1d + ReLU in PyTorch fp32."""
    import torch.nn.functional as F
    from mvp_benchmark_amd import pointwise as pw
    from mvp_benchmark_amd.pointwise import PointwiseConv1d, pointwise_conv
    torch.manual_seed(0)
    defaults = (pw.MFMA_DGRAD, pw.MFMA_WGRAD_MIN_CIN, pw.MFMA_TRAIN, pw.MFMA_WGRAD_MIN_POSITIONS)
    pw.MFMA_WGRAD_MIN_POSITIONS = 0                                                    # small batches through the MFMA weight gradient too
    for cin, cout, L, dgrad, wmin in ((128, 256, 768, False, 513), (128, 256, 768, True, 32), (64, 64, 512, True, 32),
                                      (256, 3, 300, False, 513), (24, 24, 256, False, 513), (515, 128, 384, True, 32),
                                      (1090, 256, 256, False, 513), (96, 160, 1000, True, 32), (272, 8, 768, True, 1),
                                      (3, 128, 512, True, 1), (8, 128, 768, True, 1), (68, 2, 1024, True, 1)):
        pw.MFMA_DGRAD, pw.MFMA_WGRAD_MIN_CIN, pw.MFMA_TRAIN = dgrad, wmin, True          # every route of the backward pass
        layer = PointwiseConv1d(cin, cout).to(DEV)
        x = torch.randn(4, cin, L, device=DEV, requires_grad=True)
        go = torch.randn(4, cout, L, device=DEV)
        for relu in (False, True):
            y = pointwise_conv(x, layer.weight, layer.bias, relu=relu)
            gx, gw, gb = torch.autograd.grad(y, (x, layer.weight, layer.bias), go)
            yr = conv_ref(x, layer.weight, layer.bias)
            yr = torch.relu(yr) if relu else yr
            rx, rw, rb = torch.autograd.grad(yr, (x, layer.weight, layer.bias), go)
            for a, r, name in ((y, yr, "y"), (gx, rx, "gx"), (gw, rw, "gw"), (gb, rb, "gb")):
                scale = r.abs().max().item() + 1e-6
                assert (a - r).abs().max().item() < 2e-5 * scale * math.sqrt(max(cin, L)), (cin, cout, relu, name)
    pw.MFMA_DGRAD, pw.MFMA_WGRAD_MIN_CIN, pw.MFMA_TRAIN, pw.MFMA_WGRAD_MIN_POSITIONS = defaults


@pytest.mark.gpu
@pytest.mark.parametrize("B,cin,cout,L", [(8, 512, 1024, 2048), (3, 40, 33, 20), (64, 128, 96, 384), (2, 7, 9, 16384),
                                          (5, 64, 1024, 64), (1, 3, 2, 1)])
def test_pointwise_conv_max_backward_kernels(B, cin, cout, L):
    """mvp_pointwise_max_backward (the conv -> max-over-positions backward through the B * Cout winning positions
    only) against plain autograd on conv(x).max(dim=2): same values, gradients at float32 summation-order
    tolerance, bit-reproducible; half of the columns duplicated so that maxima are attained twice (the gradient
    goes to the position torch.max reports); (2, 7, 9, 16384) exceeds the kernel's LDS budget and takes the
    PyTorch formulation of the same sparse pass."""
    from mvp_benchmark_amd.pointwise import PointwiseConv1d
    torch.manual_seed(B * 31 + L)
    layer = PointwiseConv1d(cin, cout).to(DEV)
    x = torch.randn(B, cin, L, device=DEV)
    if L >= 4:
        x[..., L // 2:L // 2 * 2] = x[..., :L // 2]
    if L >= 64:
        x[..., 5] *= 40.0                                                # a "critical point": hundreds of channels win at one position
        x[..., L // 2 + 5] = x[..., 5]
    x.requires_grad_()
    go = torch.randn(B, cout, device=DEV)
    params = (x, layer.weight, layer.bias)
    got = layer.max_over_positions(x)
    y = layer(x)
    ref, idx = y.max(dim=2)
    assert torch.equal(got, ref)
    g1 = torch.autograd.grad(got, params, go)
    g2 = torch.autograd.grad(got.clone() if False else layer.max_over_positions(x), params, go)
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)                                         # fixed summation order
    # float64 statement of the same function, gradient routed to the positions torch.max reported
    xd, wd, bd = x.detach().double(), layer.weight.detach().double().view(cout, cin), layer.bias.detach().double()
    god = go.double()
    cols = torch.gather(xd, 2, idx.unsqueeze(1).expand(B, cin, cout))    # [b, ci, co] = x[b, ci, idx[b, co]]
    want_w = torch.einsum("bo,bio->oi", god, cols)
    want_x = torch.zeros_like(xd).scatter_add_(2, idx.unsqueeze(1).expand(B, cin, cout), god.unsqueeze(1) * wd.t().unsqueeze(0))
    tol = 1e-5 * max(1.0, float(want_w.abs().max()))
    assert (g1[1].double().view(cout, cin) - want_w).abs().max().item() < tol
    assert (g1[2].double() - god.sum(0)).abs().max().item() < 1e-5 * max(1.0, float(god.sum(0).abs().max()))
    assert (g1[0].double() - want_x).abs().max().item() < 1e-5 * max(1.0, float(want_x.abs().max()))
    assert bd.shape == (cout,)


@pytest.mark.gpu
def test_pointwise_kernels_fixed_seed_fuzz_slice():
    """40 cases of tools/fuzz_pointwise.py (random batch / channel / position counts incl. 1-channel layers, ragged
    tiles, 4-position clouds, masks, padded weight rows): the MFMA forward / data gradient / weight gradient and the
    sparse conv -> max backward against float64 PyTorch.  The 300-case log: profiles/r4_fuzz_pointwise.txt."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_pointwise
    assert fuzz_pointwise.run(40, 7, verbose=False) == 0
    assert fuzz_pointwise.run_ex(30, 11, verbose=False) == 0          # the fused prologue / epilogues of ABI 18, bit for bit


@pytest.mark.gpu
@pytest.mark.parametrize("B,share,Cw,k,N", [(2, 8, 2, 20, 3072), (3, 8, 16, 10, 384), (1, 4, 3, 5, 77), (2, 1, 5, 3, 300),
                                            (2, 16, 1, 20, 1536), (64, 8, 2, 10, 3072)])
def test_share_gather_sum_equals_gather_then_weighted_sum(B, share, Cw, k, N):
    """mvp_share_gather_sum (the neighbours' values gathered and summed with their shared weights in one kernel, the
    (B, C, k, N) tensor never formed) against grouping_operation + share_weighted_sum: outputs and both gradients
    BIT-IDENTICAL (same arithmetic, same order; the gradient of v goes through the same inverted-index scatter)."""
    from mvp_benchmark_amd.mm3d_pn2.functional import grouping_operation, share_gather_sum, share_weighted_sum
    g = torch.Generator().manual_seed(B * 100 + N)
    C = share * Cw
    w = torch.randn(B, Cw, k, N, generator=g).to(DEV).requires_grad_()
    v = torch.randn(B, C, N, generator=g).to(DEV).requires_grad_()
    idx = torch.randint(0, N, (B, k, N), generator=g, dtype=torch.int32).to(DEV)
    go = torch.randn(B, C, N, generator=g).to(DEV)
    got = share_gather_sum(w, v, idx)
    ref = share_weighted_sum(w, grouping_operation(v, idx))
    assert torch.equal(got, ref)
    for a, b in zip(torch.autograd.grad(got, (w, v), go), torch.autograd.grad(ref, (w, v), go)):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("first,four_d,relu", [(True, True, True), (False, False, True), (False, False, False)])
def test_conv_interp_concat_equals_interpolate_then_convolve(first, four_d, relu, monkeypatch):
    """models/_common.py: conv_interp_concat (the interpolated half of a U-Net's up-convolution convolved at the coarse
    level, before three_interpolate) against the reference's order -- interpolate, concatenate, convolve (vrcnet.py
    :287-296, ecg.py:143-150) -- values and every gradient at float32 summation-order tolerance."""
    sys.path.insert(0, COMPLETION)
    from models._common import conv_interp_concat, pointwise1d, pointwise2d
    from model_utils import three_nn_upsampling
    torch.manual_seed(21)
    B, cc, cs, cout, nc, n = 4, 96, 40, 64, 192, 384
    conv = (pointwise2d if four_d else pointwise1d)(cc + cs, cout).to(DEV)
    pts_f, pts_c = torch.rand(B, n, 3, device=DEV), torch.rand(B, nc, 3, device=DEV)
    idx, weight = three_nn_upsampling(pts_f, pts_c)
    coarse = torch.randn(B, cc, *((1, nc) if four_d else (nc,)), device=DEV, requires_grad=True)
    skip = torch.randn(B, cs, *((1, n) if four_d else (n,)), device=DEV, requires_grad=True)
    params = (coarse, skip) + tuple(conv.parameters())
    out = []
    for ref in (False, True):
        import op_config
        monkeypatch.setattr(op_config.OPS, "conv_before_interp", not ref)
        y = conv_interp_concat(conv, coarse, skip, idx, weight, interp_first=first, relu=relu)
        out.append((y,) + torch.autograd.grad(y.square().sum(), params))
    for a, b in zip(*out):
        assert a.shape == b.shape and (a - b).abs().max().item() <= 2e-4 * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("B,cin,cout,L,relu", [(3, 512, 1024, 2048, 0), (2, 64, 48, 384, 0), (2, 130, 200, 1000, 1), (5, 32, 1024, 132, 0),
                                               (1, 256, 64, 4, 0)])
def test_pointwise_mfma_max_equals_gemm_then_max(B, cin, cout, L, relu):
    """mvp_pointwise_mfma_max (round 5: the max over a cloud's positions inside the GEMM's epilogue, pcn.py:29-30 /
    vrcnet.py:281-282) through the C ABI: values BIT-identical to mvp_pointwise_mfma's output reduced with max, positions =
    the FIRST position that attains each maximum -- also where the maximum is attained several times (duplicated
    columns) and in rows of the last, partial tiles."""
    from mvp_benchmark_amd import _lib
    g = torch.Generator().manual_seed(B * 1000 + L)
    x = torch.randn(B, cin, L, generator=g)
    x[:, :, L // 2:] = x[:, :, :L - L // 2]                 # every column occurs twice: every maximum is attained twice
    x = x.to(DEV)
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(DEV)
    ldw = 0
    if cin % 4:
        w = torch.nn.functional.pad(w, (0, -cin % 4)).contiguous()
        ldw = w.size(1)
    bias = torch.randn(cout, generator=g).to(DEV)
    y = torch.empty(B, cout, L, device=DEV)
    _lib.call("mvp_pointwise_mfma", DEV, B, cin, cout, L, x, None, w, ldw, 0, bias, None, relu, 1, y)
    val = torch.empty(B, cout, device=DEV)
    idx = torch.empty(B, cout, dtype=torch.int32, device=DEV)
    keys = torch.full((B * cout,), -1, dtype=torch.int64, device=DEV)     # (contents irrelevant)
    _lib.call("mvp_pointwise_mfma_max", DEV, B, cin, cout, L, x, w, ldw, bias, relu, val, idx, keys, keys.numel() * 8)
    torch.cuda.synchronize()
    want = y.max(dim=2)[0]
    assert torch.equal(val, want)
    first = (y == want.unsqueeze(2)).int().argmax(dim=2).int()
    assert torch.equal(idx, first)
    assert int(idx.max()) < L - L // 2 or relu                            # the first of the two copies (ReLU zeros may tie earlier still)


def test_pointwise_mfma_max_propagates_nan_and_orders_signed_zero_like_torch_max():
    """a NaN of EITHER sign in a row's products makes that row's maximum NaN at the first NaN position (torch.max;
    the key order used to rank a sign-bit NaN lowest and could leave the position -1), and -0 ties with +0 (first position)."""
    from mvp_benchmark_amd import _lib
    B, cin, cout, L = 2, 64, 96, 512
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, cin, L, generator=g)
    neg_nan = torch.tensor([0xFFC00000 - (1 << 32)], dtype=torch.int32).view(torch.float32)[0]
    x[0, :, 37] = neg_nan                                          # cloud 0: every row sees a -NaN at position 37 (and a +NaN later)
    x[0, :, 300] = float("nan")
    x[1] = 0.0                                                     # cloud 1: all products are +-0 -> every row ties everywhere
    x = x.to(DEV)
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(DEV)
    w[5] = -w[5].abs()
    bias = torch.zeros(cout, device=DEV)
    val = torch.empty(B, cout, device=DEV)
    idx = torch.empty(B, cout, dtype=torch.int32, device=DEV)
    keys = torch.zeros(B * cout, dtype=torch.int64, device=DEV)
    _lib.call("mvp_pointwise_mfma_max", DEV, B, cin, cout, L, x, w, 0, bias, 0, val, idx, keys, keys.numel() * 8)
    torch.cuda.synchronize()
    assert torch.isnan(val[0]).all() and (idx[0] == 37).all()
    assert (val[1] == 0).all() and (idx[1] == 0).all()
    y = torch.empty(B, cout, L, device=DEV)
    _lib.call("mvp_pointwise_mfma", DEV, B, cin, cout, L, x, None, w, 0, 0, bias, None, 0, 1, y)
    ref_v, ref_i = y.max(dim=2)
    assert torch.isnan(ref_v[0]).all() and torch.equal(ref_i[0].int(), idx[0]) and torch.equal(ref_v[1], val[1])


def test_conv_max_layer_uses_the_fused_forward_and_matches_autograd():
    """PointwiseConv1d.max_over_positions on an MFMA-routed shape: forward through mvp_pointwise_mfma_max, backward through
    mvp_pointwise_max_backward -- values equal conv(x).max, gradients equal autograd's of the unfused formulation."""
    from mvp_benchmark_amd import pointwise
    torch.manual_seed(3)
    layer = pointwise.PointwiseConv1d(256, 512).to(DEV)
    x = torch.randn(4, 256, 768, device=DEV, requires_grad=True)
    assert pointwise.mfma_conv_max(x.detach(), layer.weight.detach(), layer.bias.detach()) is not None
    got = layer.max_over_positions(x)
    ref = layer(x).max(dim=2)[0]
    assert torch.equal(got, ref)
    go = torch.randn_like(ref)
    params = (x,) + tuple(layer.parameters())
    for a, b in zip(torch.autograd.grad(got, params, go), torch.autograd.grad(ref, params, go)):
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item())
    with torch.no_grad():
        assert torch.equal(layer.max_over_positions(x), ref)


@pytest.mark.gpu
def test_bench_line_carries_the_contract_fields(tmp_path):
    """`python bench.py` on a small shape: ONE JSON line with the driver's contract keys, `roofline`, `cpu_baseline`, and
    (where oracle/_ref travelled) `gpu_reference_baseline` -- the reference's own kernels on the timed batch."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--batch", "4", "--points", "2048", "--iters", "200",
                          "--steps", "2", "--warmup", "1", "--cpu-sample", "2", "--no-side"],
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert abs(d["value"] - 4 * 2048 * 2048 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in d["roofline"], key
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    from oracle import ref_gpu
    g = d.get("gpu_reference_baseline")
    if ref_gpu.available(""):
        assert g and "error" not in g, g
        assert g["ms_per_step"] > 0 and g["speedup_of_this_repo"] > 1.0
        # the same batch through the reference's emd_cuda.cu and through the product: the matching cost agrees to the north
        # star's 1e-5 at this size (2048 points: the reference is deterministic here; at the headline size it differs from
        # ITSELF by ~4e-5 from run to run -- tests/test_gpu_reference_kernels.py::test_reference_emd_kernels_at_the_headline_size)
        assert abs(g["emd_mean_sqrt_dist"] - d["extra"]["metrics"]["emd"]) <= 1e-5 * d["extra"]["metrics"]["emd"], (g, d["extra"]["metrics"])
    else:
        assert g is None
