"""SURVEY 8(f) row N3 / BASELINE cfg 5: the DCP registration path (DGCNN kNN +
batched 3x3 SVD head).

CPU tests: parameter layout against the fixture generated from the imported
reference (tests/golden/make_dcp_golden.py), the closed-form SVD adjoint against
torch.linalg.svd's autograd, the metric helpers.  GPU tests: mvp_kabsch_svd3
against float64 NumPy SVDs (the oracle for this kernel), the reference-generated
forward fixture, and the cfg-5 shape (128 pairs of 1024 points)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from conftest import ROOT

REG = os.path.join(ROOT, "registration")
GOLD = os.path.join(ROOT, "tests", "golden")
if GOLD not in sys.path:
    sys.path.insert(0, GOLD)

DEV = "cuda:0"


def _reg_module(name):
    """registration/<name>.py under a private module name (completion/ has files of the same names)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("registration_" + name, os.path.join(REG, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _golden():
    return np.load(os.path.join(GOLD, "dcp_golden.npz"))


def _model():
    # registration/models, not completion/models (another test module may have imported that package)
    import importlib.util
    spec = importlib.util.spec_from_file_location("registration_dcp", os.path.join(REG, "models", "dcp.py"))
    dcp = importlib.util.module_from_spec(spec)
    saved = {k: sys.modules.pop(k) for k in ("model_utils", "train_utils") if k in sys.modules}
    sys.path.insert(0, REG)          # dcp.py's `from model_utils import ...` must find registration/'s
    try:
        spec.loader.exec_module(dcp)
    finally:
        sys.path.remove(REG)
        for k in ("model_utils", "train_utils"):
            sys.modules.pop(k, None)
        sys.modules.update(saved)
    from make_dcp_golden import fill_parameters
    net = dcp.Model(types.SimpleNamespace())
    fill_parameters(net)
    return net.eval()


def test_dcp_state_dict_layout_matches_reference():
    """Names and shapes of every parameter / buffer equal the reference model's
    (recorded from the imported reference): checkpoints interchange."""
    g = _golden()
    net = _model()
    mine = {k: str(list(v.shape)) for k, v in net.state_dict().items()}
    ref = dict(zip(g["names"].tolist(), g["shapes"].tolist()))
    assert mine == ref
    assert sum(p.numel() for p in net.parameters()) == 5568905
    import yaml
    cfg = yaml.safe_load(open(os.path.join(REG, "cfgs", "dcp.yaml")))
    assert cfg["model_name"] == "dcp" and cfg["max_angle"] == 180 and cfg["num_points"] == 2048
    assert {'batch_size', 'workers', 'nepoch', 'model_name', 'load_model', 'start_epoch', 'work_dir', 'flag',
            'manual_seed', 'step_interval_to_print', 'step_interval_to_plot', 'epoch_interval_to_save',
            'epoch_interval_to_val', 'lr', 'lr_decay', 'lr_decay_rate', 'lr_clip', 'optimizer', 'weight_decay',
            'betas', 'use_rri', 'rri_size', 'num_clusters', 'num_points', 'use_tnet', 'use_fpfh', 'use_ppf',
            'descriptor_size', 'max_angle', 'max_trans', 'category', 'benchmark', 'num_rot_levels',
            'num_corr_levels'} == set(cfg)


def test_svd_adjoint_matches_autograd():
    """svd3_kabsch_backward == autograd through torch.linalg.svd of
    R = V diag(1,1,d) U^T with the reflection d held fixed (float64)."""
    from mvp_benchmark_amd.registration import svd3_kabsch_backward
    torch.manual_seed(0)
    H = torch.randn(32, 3, 3, dtype=torch.float64, requires_grad=True)
    U, S, Vh = torch.linalg.svd(H)
    V = Vh.transpose(1, 2)
    flip = torch.linalg.det(V @ U.transpose(1, 2)) < 0
    assert 4 < int(flip.sum()) < 28
    d = torch.ones(32, 3, dtype=torch.float64)
    d[:, 2] = torch.where(flip, -1.0, 1.0)
    R = (V * d.unsqueeze(1)) @ U.transpose(1, 2)
    g = torch.randn(32, 3, 3, dtype=torch.float64)
    want, = torch.autograd.grad((R * g).sum(), H)
    got = svd3_kabsch_backward(U.detach(), S.detach(), V.detach(), flip.int(), g)
    assert torch.allclose(got, want, rtol=1e-10, atol=1e-10)


def test_metric_helpers():
    tu = _reg_module("train_utils")
    ang = torch.tensor([0.3, 1.2])
    c, s = torch.cos(ang), torch.sin(ang)
    z, o = torch.zeros(2), torch.ones(2)
    R = torch.stack([c, -s, z, s, c, z, z, z, o], dim=1).view(2, 3, 3)
    eye = torch.eye(3).expand(2, 3, 3)
    assert torch.allclose(tu.rotation_error(R, eye), ang * 180 / np.pi, atol=1e-3)
    assert torch.allclose(tu.rotation_geodesic_error(R, eye), ang, atol=1e-5)
    t = torch.tensor([[1.0, 2.0, 2.0], [0.0, 0.0, 0.0]])
    T = tu.rt_to_transformation(R, t.unsqueeze(2))
    assert T.shape == (2, 4, 4) and torch.equal(T[:, 3], torch.tensor([0.0, 0.0, 0.0, 1.0]).expand(2, 4))
    assert torch.allclose(tu.translation_error(t, torch.zeros(2, 3)), torch.tensor([3.0, 0.0]))
    pts = torch.rand(2, 5, 3)
    assert torch.allclose(tu.rmse_loss(pts, T, T), torch.zeros(2), atol=1e-7)
    q = torch.tensor([[0.0, 0.0, np.sin(0.15), np.cos(0.15)]], dtype=torch.float32)   # (x, y, z, w): 0.3 rad about z
    assert torch.allclose(tu.quat2mat(q)[0], R[0], atol=1e-6)


def _np_kabsch(H):
    """float64 NumPy restatement of SVDHead's loop (dcp.py:360-368)."""
    out, flips = [], []
    for h in H.astype(np.float64):
        u, s, vt = np.linalg.svd(h)
        v = vt.T
        r = v @ u.T
        flip = np.linalg.det(r) < 0
        if flip:
            v = v @ np.diag([1.0, 1.0, -1.0])
            r = v @ u.T
        out.append(r)
        flips.append(flip)
    return np.stack(out), np.array(flips)


@pytest.mark.gpu
def test_kabsch_svd3_matches_float64_numpy():
    from mvp_benchmark_amd.registration import svd3
    rng = np.random.default_rng(0)
    H = rng.standard_normal((128, 3, 3)).astype(np.float32)
    H[:8] *= 1e-3                                     # small correlations (nearly identical clouds)
    H[8:16] *= 1e3
    U, S, V, R, flipped = [t.cpu().numpy() for t in svd3(torch.tensor(H, device=DEV))]
    want, flips = _np_kabsch(H)
    assert 30 < flips.sum() < 100                     # both branches of the reflection fix
    np.testing.assert_array_equal(flipped.astype(bool), flips)
    np.testing.assert_allclose(R, want, atol=2e-6)
    np.testing.assert_allclose(np.linalg.det(R.astype(np.float64)), 1.0, atol=1e-5)
    # factors: H = U diag(S) V^T, S descending, U and V orthogonal
    rec = np.einsum("bij,bj,bkj->bik", U.astype(np.float64), S.astype(np.float64), V.astype(np.float64))
    scale = np.abs(H).max(axis=(1, 2), keepdims=True).astype(np.float64)
    np.testing.assert_allclose(rec / scale, H / scale, rtol=0, atol=2e-6)
    assert (S[:, 0] >= S[:, 1]).all() and (S[:, 1] >= S[:, 2]).all() and (S >= 0).all()
    np.testing.assert_allclose(S, np.linalg.svd(H.astype(np.float64), compute_uv=False), rtol=2e-6)
    eye = np.eye(3)
    np.testing.assert_allclose(np.einsum("bji,bjk->bik", U, U), np.broadcast_to(eye, U.shape), atol=2e-6)
    np.testing.assert_allclose(np.einsum("bji,bjk->bik", V, V), np.broadcast_to(eye, V.shape), atol=2e-6)


@pytest.mark.gpu
def test_kabsch_svd3_degenerate_inputs():
    """Planar clouds (rank 2): the rotation after the reflection fix is unique and
    must match NumPy; rank 1 / 0: still a proper rotation, nothing NaN."""
    from mvp_benchmark_amd.registration import svd3
    rng = np.random.default_rng(1)
    A = rng.standard_normal((16, 3, 2))
    B = rng.standard_normal((16, 2, 3))
    H2 = (A @ B).astype(np.float32)                   # rank 2
    a = rng.standard_normal((8, 3, 1))
    H1 = (a @ rng.standard_normal((8, 1, 3))).astype(np.float32)   # rank 1
    H0 = np.zeros((2, 3, 3), np.float32)
    Hs = np.concatenate([H2, H1, H0])
    _, S, _, R, _ = [t.cpu().numpy() for t in svd3(torch.tensor(Hs, device=DEV))]
    assert np.isfinite(R).all() and np.isfinite(S).all()
    np.testing.assert_allclose(np.einsum("bji,bjk->bik", R, R), np.broadcast_to(np.eye(3), R.shape), atol=5e-6)
    np.testing.assert_allclose(np.linalg.det(R.astype(np.float64)), 1.0, atol=1e-5)
    want, _ = _np_kabsch(H2)
    np.testing.assert_allclose(R[:16], want, atol=5e-5)
    np.testing.assert_array_equal(R[-2:], np.broadcast_to(np.eye(3, dtype=np.float32), (2, 3, 3)))


@pytest.mark.gpu
def test_kabsch_rotation_gradient():
    from mvp_benchmark_amd.registration import kabsch_rotation
    torch.manual_seed(3)
    H = torch.randn(64, 3, 3)
    g = torch.randn(64, 3, 3)
    Hd = H.to(DEV).requires_grad_()
    R = kabsch_rotation(Hd)
    (R * g.to(DEV)).sum().backward()
    H64 = H.double().requires_grad_()
    U, S, Vh = torch.linalg.svd(H64)
    V = Vh.transpose(1, 2)
    d = torch.ones(64, 3, dtype=torch.float64)
    d[:, 2] = torch.where(torch.linalg.det(V @ U.transpose(1, 2)) < 0, -1.0, 1.0)
    R64 = (V * d.unsqueeze(1)) @ U.transpose(1, 2)
    want, = torch.autograd.grad((R64 * g.double()).sum(), H64)
    well = (S[:, 0] - S[:, 1] > 0.05) & (S[:, 1] - S[:, 2] > 0.05)          # away from the adjoint's poles
    assert int(well.sum()) > 40
    np.testing.assert_allclose(R.detach().cpu().numpy(), R64.detach().float().numpy(), atol=2e-6)
    np.testing.assert_allclose(Hd.grad.cpu().numpy()[well.numpy()], want.float().numpy()[well.numpy()], rtol=2e-3, atol=2e-3)


@pytest.mark.gpu
def test_dcp_forward_matches_reference_fixture():
    """Same (closed-form) parameters as the imported reference, its input, eval
    mode: T_12 and the evaluation metrics reproduce the reference's outputs."""
    g = _golden()
    net = _model().to(DEV)
    src, tgt, T_gt = (torch.tensor(g[k], device=DEV) for k in ("src", "tgt", "T_gt"))
    with torch.no_grad():
        T_12 = net(src, tgt)
        loss, r_err, t_err, rmse, rt_mse = net(src, tgt, T_gt)
    np.testing.assert_allclose(T_12.cpu().numpy(), g["T_12"], atol=2e-4)
    np.testing.assert_allclose(loss.cpu().numpy(), g["loss"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(r_err.cpu().numpy(), g["r_err"], rtol=1e-3, atol=2e-2)
    np.testing.assert_allclose(t_err.cpu().numpy(), g["t_err"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(rmse.cpu().numpy(), g["rmse"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(rt_mse.cpu().numpy(), g["rt_mse"], rtol=1e-3, atol=5e-4)


@pytest.mark.gpu
def test_dcp_graph_feature_uses_the_op_layer_and_matches_the_matmul_formulation():
    mu = _reg_module("model_utils")
    g = torch.Generator().manual_seed(2)
    x = torch.rand(4, 3, 1024, generator=g)
    got = mu.get_graph_feature(x.to(DEV), k=20).cpu()
    want = mu.get_graph_feature(x.double(), k=20).float()          # reference formulation (CPU, float64)
    assert got.shape == (4, 6, 1024, 20)
    # same neighbour sets (order inside a neighbourhood may differ only between equal distances)
    np.testing.assert_allclose(np.sort(got[:, :3].numpy(), axis=-1), np.sort(want[:, :3].numpy(), axis=-1), atol=1e-6)
    assert torch.equal(got[:, 3:], want[:, 3:])


@pytest.mark.gpu
def test_dcp_cfg5_shape_runs():
    """BASELINE cfg 5: 128 pairs of 1024 points, kNN + SVD path, one MI355X.
    tgt is src moved by a known pose; outputs are proper rigid motions."""
    tu = _reg_module("train_utils")
    net = _model().to(DEV)
    g = torch.Generator().manual_seed(5)
    src = (torch.rand(128, 1024, 3, generator=g) - 0.5).to(DEV)
    q = torch.nn.functional.normalize(torch.randn(128, 4, generator=g), dim=1).to(DEV)
    Rg = tu.quat2mat(q)
    tg = (torch.rand(128, 3, generator=g) - 0.5).to(DEV)
    tgt = src @ Rg.transpose(1, 2) + tg.unsqueeze(1)
    T_gt = tu.rt_to_transformation(Rg, tg.unsqueeze(2))
    with torch.no_grad():
        loss, r_err, t_err, rmse, rt_mse = net(src, tgt, T_gt)
        T = net(src, tgt)
    assert T.shape == (128, 4, 4) and torch.isfinite(T).all()
    R = T[:, :3, :3]
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3, device=DEV).expand(128, 3, 3), atol=1e-5)
    assert torch.allclose(torch.linalg.det(R), torch.ones(128, device=DEV), atol=1e-5)
    for v in (loss, r_err, t_err, rmse, rt_mse):
        assert torch.isfinite(v).all()
    assert r_err.shape == (128,) and float(r_err.max()) <= 180.0
