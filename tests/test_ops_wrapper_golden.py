"""The Python operator layer (mvp_benchmark_amd/mm3d_pn2: functional.py wrappers and modules.py) against fixtures generated
by running the REFERENCE's own wrappers (tests/golden/make_ops_golden.py: the reference's ops/<op>/<op>.py with their
compiled `*_ext` modules replaced by launcher-signature stubs on the CPU oracle).  What is pinned is what the wrappers add
to the kernels: argument order, allocation / initial values, three_nn's sqrt, knn's transposes, QueryAndGroup's composition
(centring, normalisation, uniform_sample's draws under a seed), GroupAll, Points_Sampler's ranges and offsets, F-FPS's
distance matrix.

"cpu": the modules' Python with the operator names patched by the oracle-backed callables (the product's own raise
without a GPU); "cuda" (-m gpu): the wrappers and kernels end to end -- exact equality (the kernels are bit-identical to
the oracle the fixtures were computed with)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

G = np.load(os.path.join(ROOT, "tests", "golden", "ops_wrapper_golden.npz"))


def g(key):
    return G[key]


@pytest.fixture(params=["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def ops(request, monkeypatch):
    """-> (namespace with the operator callables and module classes under test, device)."""
    import types
    from mvp_benchmark_amd.mm3d_pn2 import functional, modules
    ns = types.SimpleNamespace(QueryAndGroup=modules.QueryAndGroup, GroupAll=modules.GroupAll, Points_Sampler=modules.Points_Sampler,
                               calc_square_dist=modules.calc_square_dist)
    names = ("ball_query", "knn", "three_nn", "three_interpolate", "gather_points", "grouping_operation",
             "furthest_point_sample", "furthest_point_sample_with_dist")
    if request.param == "cpu":
        import oracle_ops
        table = dict(ball_query=oracle_ops.ball_query, knn=oracle_ops.knn_op, three_nn=oracle_ops.three_nn,
                     three_interpolate=oracle_ops.three_interpolate, gather_points=oracle_ops.gather_points,
                     grouping_operation=oracle_ops.grouping_operation, furthest_point_sample=oracle_ops.furthest_point_sample,
                     furthest_point_sample_with_dist=oracle_ops.furthest_point_sample_with_dist)
        for n in names:
            if hasattr(modules, n):
                monkeypatch.setattr(modules, n, table[n])
            setattr(ns, n, table[n])
    else:
        assert torch.cuda.is_available()
        for n in names:
            setattr(ns, n, getattr(functional, n))
    return ns, torch.device(request.param)


def T(a, dev):
    return torch.from_numpy(np.asarray(a)).to(dev)


def eq(got, want):
    got = got.detach().cpu().numpy()
    assert got.shape == want.shape and got.dtype == want.dtype, (got.shape, want.shape, got.dtype, want.dtype)
    np.testing.assert_array_equal(got, want)


def _inputs(dev):
    return T(g("in/xyz"), dev), T(g("in/ctr"), dev), T(g("in/feat"), dev)


def test_query_wrappers_match_reference(ops):
    """ball_query.py:7-47 (argument order min, max, nsample, xyz, centres; zero-initialised idx), knn.py:7-72 (idx (B,k,M);
    centres default to the points; transposed inputs), three_nn.py:8-45 (the sqrt of dist2)."""
    ns, dev = ops
    xyz, ctr, _ = _inputs(dev)
    eq(ns.ball_query(0.0, 0.2, 8, xyz, ctr), g("ball_query/full"))
    eq(ns.ball_query(0.1, 0.25, 5, xyz, ctr), g("ball_query/ring"))
    eq(ns.knn(5, xyz, ctr), g("knn/centres"))
    eq(ns.knn(5, xyz), g("knn/self5"))
    eq(ns.knn(4, xyz.transpose(1, 2).contiguous(), ctr.transpose(1, 2).contiguous(), True), g("knn/transposed"))
    dist, idx = ns.three_nn(xyz, ctr)
    eq(idx, g("three/idx"))
    # the fixture's sqrt is this container's torch CPU sqrt, which is NOT correctly rounded (0.6 % of float32 inputs are
    # one ulp off the IEEE result numpy, the oracle and the GPU give): one ulp allowed
    np.testing.assert_allclose(dist.cpu().numpy(), g("three/dist"), rtol=1.3e-7, atol=0)


def test_gather_group_interpolate_wrappers_match_reference(ops):
    """three_interpolate.py:8-63, gather_points.py:7-52, group_points.py:166-221: outputs and the gradients of their
    backward() (the reference's float atomics add in an unspecified order: 1e-6)."""
    ns, dev = ops
    _, _, feat = _inputs(dev)
    cf = T(g("three/cfeat"), dev).requires_grad_()
    y = ns.three_interpolate(cf, T(g("three/idx"), dev), T(g("three/weight"), dev))
    eq(y, g("three/out"))
    y.backward(T(g("three/gy"), dev))
    np.testing.assert_allclose(cf.grad.cpu().numpy(), g("three/grad"), rtol=1e-6, atol=1e-6)
    for case, fn in (("gather", ns.gather_points), ("group", ns.grouping_operation)):
        f = feat.clone().requires_grad_()
        y = fn(f, T(g(case + "/idx"), dev))
        eq(y, g(case + "/out"))
        y.backward(T(g(case + "/gy"), dev))
        np.testing.assert_allclose(f.grad.cpu().numpy(), g(case + "/grad"), rtol=1e-6, atol=1e-6)


def test_fps_wrappers_and_square_dist_match_reference(ops):
    """furthest_point_sample.py:7-78 (temp pre-filled with 1e10), utils.py:4-31 calc_square_dist (the float32 order
    |a|^2 + |b|^2 - 2ab; GPU: another matmul accumulation -> 1e-6)."""
    ns, dev = ops
    xyz, ctr, _ = _inputs(dev)
    eq(ns.furthest_point_sample(xyz, 33), g("fps/idx"))
    eq(ns.furthest_point_sample_with_dist(T(g("fps/dmat"), dev), 20), g("fps/with_dist"))
    xs = xyz[:, :64].contiguous()
    tol = dict(rtol=0, atol=0) if dev.type == "cpu" else dict(rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(ns.calc_square_dist(xs, xs, norm=False).cpu().numpy(), g("fps/dmat"), **tol)
    tol = dict(rtol=0, atol=0) if dev.type == "cpu" else dict(rtol=1e-3, atol=1e-4)     # (sqrt of ~0 on the diagonal-free matrix: loose only there)
    np.testing.assert_allclose(ns.calc_square_dist(xs, ctr, norm=True).cpu().numpy(), g("fps/dnorm"), **tol)


@pytest.mark.parametrize("name,kw", [
    ("ball", dict(max_radius=0.2, sample_num=8)),
    ("ball_norm_xyz", dict(max_radius=0.2, sample_num=8, min_radius=0.05, normalize_xyz=True, return_grouped_xyz=True)),
    ("knn", dict(max_radius=None, sample_num=6)),
    ("no_xyz", dict(max_radius=0.25, sample_num=4, use_xyz=False)),
    ("uniform", dict(max_radius=0.12, sample_num=8, uniform_sample=True, return_unique_cnt=True)),
])
def test_query_and_group_matches_reference(ops, name, kw):
    """group_points.py:11-122, every switch; uniform_sample under the generator's seed (its draws come from the HOST
    generator in the reference: group_points.py:86-90)."""
    ns, dev = ops
    xyz, ctr, feat = _inputs(dev)
    torch.manual_seed(7)
    r = ns.QueryAndGroup(**kw)(xyz, ctr, feat)
    r = r if isinstance(r, tuple) else (r,)
    for i, t in enumerate(r):
        if kw.get("normalize_xyz") and dev.type == "cuda":
            # `grouped_xyz /= max_radius`: PyTorch divides by a Python scalar as a multiplication by its reciprocal on the GPU
            # (the reference on CUDA does the same) and as a true division on the CPU, where the fixture was computed: one ulp
            np.testing.assert_allclose(t.cpu().numpy(), g("qag_%s/out%d" % (name, i)), rtol=2.5e-7, atol=1e-9)
            continue
        eq(t, g("qag_%s/out%d" % (name, i)))


def test_group_all_xyz_only_and_samplers_match_reference(ops):
    """group_points.py:125-163 GroupAll; QueryAndGroup without features; points_sampler.py:34-158 Points_Sampler: D-FPS,
    F-FPS and FS over index ranges as the reference slices them (`last_fps_end_index += fps_sample_range`), offsets added."""
    ns, dev = ops
    xyz, ctr, feat = _inputs(dev)
    eq(ns.QueryAndGroup(0.2, 8)(xyz, ctr, None), g("qag_xyz_only/out0"))
    eq(ns.GroupAll(True)(xyz, ctr, feat), g("group_all/with_feat"))
    eq(ns.GroupAll(False)(xyz, ctr, feat), g("group_all/feat_only"))
    eq(ns.GroupAll(True)(xyz, ctr, None), g("group_all/xyz_only"))
    for name, (num, mods, ranges) in {"dfps": ([16], ["D-FPS"], [-1]), "mixed": ([8, 6, 5], ["D-FPS", "F-FPS", "FS"], [100, 120, -1]),
                                     "ffps": ([12], ["F-FPS"], [-1])}.items():
        got = ns.Points_Sampler(num, mods, ranges)(xyz, feat)
        if dev.type == "cuda" and name != "dfps":
            # F-FPS ranks by a float32 distance MATRIX (|a|^2 + |b|^2 - 2ab): the GPU's matmul rounds it differently, and a
            # near-tie of the running maximum may branch -- compared on the CPU exactly, here by shape / range only
            want = g("sampler_%s/idx" % name)
            assert tuple(got.shape) == want.shape and int(got.min()) >= 0 and int(got.max()) < xyz.size(1)
            continue
        eq(got, g("sampler_%s/idx" % name))


@pytest.mark.gpu
def test_metric_modules_match_reference_wrappers():
    """cd() / emd() (mvp_benchmark_amd/metrics) against the reference's own chamfer_3DDist / emdModule run around
    launcher stubs on the oracle (dist_chamfer_3D.py:26-74, emd_module.py:40-88): forward values and indices exact, the
    gradients of backward() (float atomics in the reference: 1e-6), no gradient to emd's second input."""
    from mvp_benchmark_amd.metrics import cd, emd
    dev = torch.device("cuda")
    a, b = T(g("cd/a"), dev).requires_grad_(), T(g("cd/b"), dev).requires_grad_()
    d1, d2, i1, i2 = cd()(a, b)
    for t, k in ((d1, "dist1"), (d2, "dist2"), (i1, "idx1"), (i2, "idx2")):
        eq(t, g("cd/" + k))
    torch.autograd.backward([d1, d2], [T(g("cd/g1"), dev), T(g("cd/g2"), dev)])
    np.testing.assert_allclose(a.grad.cpu().numpy(), g("cd/grad_a"), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(b.grad.cpu().numpy(), g("cd/grad_b"), rtol=1e-5, atol=1e-6)
    p, q = T(g("emd_in/p"), dev).requires_grad_(), T(g("emd_in/q"), dev).requires_grad_()
    for tag, eps, iters in (("train", 0.005, 50), ("smoke", 0.05, 3000)):
        p.grad = None
        dist, ass = emd()(p, q, eps, iters)
        eq(dist, g("emd_%s/dist" % tag))
        eq(ass, g("emd_%s/assignment" % tag))
        dist.backward(T(g("emd_%s/gd" % tag), dev))
        np.testing.assert_allclose(p.grad.cpu().numpy(), g("emd_%s/grad_p" % tag), rtol=1e-6, atol=1e-7)
        assert q.grad is None or float(q.grad.abs().max()) == 0.0
        # the reference's own smoke check (emd_module.py:100-104): dist = |x1 - x2[assignment]|^2
        x2 = q.detach().gather(1, ass.long().unsqueeze(2).expand(-1, -1, 3))
        np.testing.assert_allclose(((p.detach() - x2) ** 2).sum(2).cpu().numpy(), dist.detach().cpu().numpy(), rtol=1e-5, atol=1e-7)
