"""The REFERENCE's own kernels, run on the MI355X, against the CPU oracle and against this repo's HIP path.

oracle/build_ref_gpu.sh compiles the reference's .cu files for gfx950 from where they lie (hipify-perl + hipcc, the image's
own tools; .so files only, in oracle/_ref/, which travels to the GPU box).  Two builds:

  ""        hipcc's default floating-point contraction (the counterpart of nvcc's default -fmad=true).  WHICH products
            get fused is the compiler's choice (hipcc's differs from nvcc's and varies from kernel to kernel), so values
            may differ from the oracle's canonical chain in the last place: indices must be IDENTICAL, values within
            `ULP` = 2.5e-7 relative (north_star asks 1e-5).
  "_nofma"  -ffp-contract=off: the arithmetic of the .cu text as written.  Against the oracle in the same mode
            (oracle.set_contraction(False)) EVERYTHING must be bit-identical, values included -- this pins the oracle's
            algorithm (control flow, tie rules, thread partitions, initial values) to the reference's kernels exactly, and
            leaves the spelling of one sum of three products as the only difference between oracle modes.

Float-atomic gradients (gather / group / interpolate / chamfer backward) depend on the order the hardware performs the
adds: rtol 1e-5 as everywhere else.  EMD's GetMax (emd_cuda.cu:181-194) lets the LAST writer among near-tied bidders win:
where the oracle's result does not depend on that choice (lowest- and highest-index policies agree) the reference's
kernels must give the oracle's assignment exactly; where it does, the reference differs from ITSELF from run to run and
only the matching cost is compared (5e-3: the reference's own runs at (2,8192) are spread over 2.3e-3,
profiles/r5_reference_kernels.txt; profiles/r2_emd_schedule_sensitivity.txt).
"""
import numpy as np
import pytest
import torch

import ref_kernels as ref
from conftest import rand_clouds

pytestmark = pytest.mark.gpu

ULP = 2.5e-7
VARIANTS = ["", "_nofma"]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


@pytest.fixture(params=VARIANTS, ids=["default-contraction", "no-contraction"])
def mode(request, oracle):
    """(variant, exact): the oracle is put into the arithmetic the variant was compiled with and restored afterwards."""
    v = request.param
    if not ref.available(v):
        pytest.skip("oracle/_ref is not built (oracle/build_ref_gpu.sh needs /root/reference; run build() in the container)")
    oracle.set_contraction(v != "_nofma")
    yield v, v == "_nofma"
    oracle.set_contraction(True)


def same_index(a, b):
    np.testing.assert_array_equal(host(a), host(b))


def same_value(a, b, exact):
    a, b = host(a), host(b)
    if exact:
        np.testing.assert_array_equal(a, b)
    else:
        np.testing.assert_allclose(a, b, rtol=ULP, atol=0)


@pytest.mark.parametrize("b,n,m", [(4, 2048, 512), (3, 1000, 300), (2, 16384, 2048), (5, 100, 37), (2, 8193, 1024), (2, 513, 200),
                                   (1, 4096, 4096)])
def test_reference_fps_kernel(oracle, mode, b, n, m):
    from mvp_benchmark_amd import mm3d_pn2 as pn2
    v, _ = mode
    x = rand_clouds(n + m, b, n, 3)
    r = ref.fps(dev(x), m, v)
    same_index(r, oracle.furthest_point_sample(x, m))
    same_index(pn2.furthest_point_sample(dev(x), m), r)


def test_reference_fps_kernel_ties_and_dist_variant(oracle, mode):
    from mvp_benchmark_amd import mm3d_pn2 as pn2
    v, _ = mode
    g = np.stack(np.meshgrid(*[np.arange(8, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(1, -1, 3) / 8
    g = np.concatenate([g, g[:, ::-1]], 0).copy()          # every distance tied many times over
    r = ref.fps(dev(g), 200, v)
    same_index(r, oracle.furthest_point_sample(g, 200))
    same_index(pn2.furthest_point_sample(dev(g), 200), r)
    x = rand_clouds(5, 3, 700, 3)
    d = ((x[:, :, None, :] - x[:, None, :, :]) ** 2).sum(-1).astype(np.float32)
    r = ref.fps_with_dist(dev(d), 128, v)
    same_index(r, oracle.furthest_point_sample_with_dist(d, 128))
    same_index(pn2.furthest_point_sample_with_dist(dev(d), 128), r)


def test_reference_query_kernels(oracle, mode):
    from mvp_benchmark_amd import mm3d_pn2 as pn2
    v, exact = mode
    xyz, ctr = rand_clouds(1, 4, 2048, 3), rand_clouds(2, 4, 512, 3)
    for lo, hi, s in [(0.0, 0.2, 32), (0.05, 0.3, 16), (0.0, 0.05, 8)]:
        r = ref.ball_query(lo, hi, s, dev(xyz), dev(ctr), v)
        same_index(r, oracle.ball_query(lo, hi, s, xyz, ctr))
        same_index(pn2.ball_query(lo, hi, s, dev(xyz), dev(ctr)), r)
    for k in (1, 8, 16, 20):
        ri, rd = ref.knn(k, dev(xyz), dev(ctr), v)
        oi, od = oracle.knn(k, xyz, ctr, return_dist=True)
        same_index(ri.transpose(2, 1), oi)
        same_value(rd, od, exact)
        same_index(pn2.knn(k, dev(xyz), dev(ctr)), ri.transpose(2, 1))
    rd, ri = ref.three_nn(dev(ctr), dev(xyz), v)
    od, oi = oracle.three_nn(ctr, xyz)
    same_index(ri, oi)
    same_value(np.sqrt(host(rd)), od, exact)
    md, mi = pn2.three_nn(dev(ctr), dev(xyz))
    same_index(mi, ri)
    same_value(md, np.sqrt(host(rd)), False)


def test_reference_query_kernels_on_a_lattice(oracle, mode):
    """Exact distance ties: the reference's heap / strict-< / first-S mechanics decide, and must decide alike."""
    from mvp_benchmark_amd import mm3d_pn2 as pn2
    v, _ = mode
    g = np.stack(np.meshgrid(*[np.arange(8, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(1, -1, 3) / 8
    xyz = np.concatenate([g, g[:, ::-1]], 0).copy()
    ctr = xyz[:, ::5].copy()
    r = ref.ball_query(0.0, 0.3, 16, dev(xyz), dev(ctr), v)
    same_index(r, oracle.ball_query(0.0, 0.3, 16, xyz, ctr))
    same_index(pn2.ball_query(0.0, 0.3, 16, dev(xyz), dev(ctr)), r)
    ri, _ = ref.knn(9, dev(xyz), dev(ctr), v)
    same_index(ri.transpose(2, 1), oracle.knn(9, xyz, ctr))
    same_index(pn2.knn(9, dev(xyz), dev(ctr)), ri.transpose(2, 1))
    _, ti = ref.three_nn(dev(ctr), dev(xyz), v)
    same_index(ti, oracle.three_nn(ctr, xyz)[1])
    same_index(pn2.three_nn(dev(ctr), dev(xyz))[1], ti)


def test_reference_gather_group_interpolate_kernels(oracle, mode):
    from mvp_benchmark_amd import mm3d_pn2 as pn2
    v, exact = mode
    rng = np.random.default_rng(4)
    feat = rand_clouds(3, 4, 24, 2048)
    idx = rng.integers(0, 2048, (4, 300)).astype(np.int32)
    r = ref.gather_points(dev(feat), dev(idx), v)
    same_value(r, oracle.gather_points(feat, idx), True)
    same_value(pn2.gather_points(dev(feat), dev(idx)), r, True)
    gidx = rng.integers(0, 2048, (4, 128, 16)).astype(np.int32)
    r = ref.grouping_operation(dev(feat), dev(gidx), v)
    same_value(r, oracle.grouping_operation(feat, gidx), True)
    same_value(pn2.grouping_operation(dev(feat), dev(gidx)), r, True)
    tidx = rng.integers(0, 2048, (4, 512, 3)).astype(np.int32)
    w = rand_clouds(8, 4, 512, 3)
    w /= w.sum(-1, keepdims=True)
    r = ref.three_interpolate(dev(feat), dev(tidx), dev(w), v)
    same_value(r, oracle.three_interpolate(feat, tidx, w), exact)
    same_value(pn2.three_interpolate(dev(feat), dev(tidx), dev(w)), r, False)
    # gradients: float atomics in the reference, hardware order
    go = rand_clouds(5, 4, 24, 300)
    r = ref.gather_points_grad(dev(go), dev(idx), 2048, v)
    np.testing.assert_allclose(host(r), oracle.gather_points_grad(go, idx, 2048), rtol=1e-5, atol=1e-6)
    f = dev(feat).requires_grad_(True)
    pn2.gather_points(f, dev(idx)).backward(dev(go))
    np.testing.assert_allclose(host(f.grad), host(r), rtol=1e-5, atol=1e-6)
    gg = rand_clouds(7, 4, 24, 128, 16)
    r = ref.grouping_operation_grad(dev(gg), dev(gidx), 2048, v)
    np.testing.assert_allclose(host(r), oracle.grouping_operation_grad(gg, gidx, 2048), rtol=1e-5, atol=1e-6)
    f = dev(feat).requires_grad_(True)
    pn2.grouping_operation(f, dev(gidx)).backward(dev(gg))
    np.testing.assert_allclose(host(f.grad), host(r), rtol=1e-5, atol=1e-6)
    gi = rand_clouds(10, 4, 24, 512)
    r = ref.three_interpolate_grad(dev(gi), dev(tidx), dev(w), 2048, v)
    np.testing.assert_allclose(host(r), oracle.three_interpolate_grad(gi, tidx, w, 2048), rtol=1e-5, atol=1e-6)
    f = dev(feat).requires_grad_(True)
    pn2.three_interpolate(f, dev(tidx), dev(w)).backward(dev(gi))
    np.testing.assert_allclose(host(f.grad), host(r), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("b,n,m", [(4, 100, 200), (2, 2048, 2048), (2, 2048, 16384), (3, 777, 1300), (1, 16384, 16384), (2, 1, 33)])
def test_reference_chamfer_kernel(oracle, mode, b, n, m):
    from mvp_benchmark_amd import metrics
    v, exact = mode
    a, c = rand_clouds(n, b, n, 3), rand_clouds(m + 1, b, m, 3)
    d1, d2, i1, i2 = ref.chamfer_forward(dev(a), dev(c), v)
    o1, o2, j1, j2 = oracle.chamfer_forward(a, c)
    same_index(i1, j1)
    same_index(i2, j2)
    same_value(d1, o1, exact)
    same_value(d2, o2, exact)
    m1, m2, k1, k2 = metrics.cd()(dev(a), dev(c))
    same_index(k1, i1)
    same_index(k2, i2)
    same_value(m1, d1, False)
    same_value(m2, d2, False)
    if n > 1 and n * m <= 2048 * 2048:
        g1, g2 = rand_clouds(11, b, n), rand_clouds(12, b, m)
        gx1, gx2 = ref.chamfer_backward(dev(a), dev(c), dev(g1), dev(g2), i1, i2, v)
        q1, q2 = oracle.chamfer_backward(a, c, g1, g2, host(i1), host(i2))
        np.testing.assert_allclose(host(gx1), q1, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(host(gx2), q2, rtol=1e-5, atol=1e-6)
        pa, pc = dev(a).requires_grad_(True), dev(c).requires_grad_(True)
        p1, p2, _, _ = metrics.cd()(pa, pc)
        (p1 * dev(g1)).sum().add((p2 * dev(g2)).sum()).backward()
        np.testing.assert_allclose(host(pa.grad), host(gx1), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(host(pc.grad), host(gx2), rtol=1e-5, atol=1e-6)


def test_reference_chamfer_kernel_exact_ties(oracle, mode):
    from mvp_benchmark_amd import metrics
    v, _ = mode
    g = np.stack(np.meshgrid(*[np.arange(6, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(1, -1, 3) / 4
    a = np.concatenate([g, g], 1).copy()                    # duplicated points: lowest index must win
    c = (g + np.float32(0.125)).copy()
    d1, d2, i1, i2 = ref.chamfer_forward(dev(a), dev(c), v)
    o = oracle.chamfer_forward(a, c)
    same_index(i1, o[2])
    same_index(i2, o[3])
    m = metrics.cd()(dev(a), dev(c))
    same_index(m[2], i1)
    same_index(m[3], i2)


EMD_CASES = [(2, 1024, 0.005, 50), (2, 1024, 0.002, 10000), (4, 2048, 0.004, 3000), (2, 2048, 0.05, 100), (3, 1024, 0.004, 3000),
             # the forced last round with most persons still unassigned (emd_cuda.cu:200); three blocks per cloud; eps extremes
             (2, 1024, 0.005, 1), (2, 1024, 0.005, 2), (2, 1024, 0.005, 5), (2, 3072, 0.01, 300), (3, 2048, 0.05, 3000),
             (2, 1024, 1e-4, 500),
             (2, 4096, 0.004, 3000), (2, 8192, 0.004, 3000)]


@pytest.mark.parametrize("b,n,eps,iters", EMD_CASES)
def test_reference_emd_kernels(oracle, mode, b, n, eps, iters):
    from mvp_benchmark_amd import metrics
    v, exact = mode
    a, c = rand_clouds(n + iters, b, n, 3), rand_clouds(n + iters + 1, b, n, 3)
    rd, ra, _ = ref.emd_forward(dev(a), dev(c), eps, iters, v)
    od, oa = oracle.emd_forward(a, c, eps, iters)
    lo = oracle.emd_forward_ex(a, c, eps, iters, getmax_lowest=True)
    policy_free = np.array_equal(np.asarray(lo[1]), oa)
    # the product computes in the canonical arithmetic whatever mode the oracle is in
    md, ma = metrics.emd()(dev(a), dev(c), eps, iters)
    assert n > 2048 or policy_free, "small cases are meant to be free of GetMax ties"
    if policy_free:
        same_index(ra, oa)
        same_value(rd, od, exact)
        if not exact:
            same_index(ma, ra)          # canonical product vs the default-contraction reference: same assignment
            same_value(md, rd, False)
    else:
        # the reference is its own moving target here (module docstring): compare the matching cost
        cost = lambda d: float(np.sqrt(host(d)).mean())
        assert abs(cost(rd) - cost(od)) <= 5e-3 * cost(od)
        assert abs(cost(rd) - cost(md)) <= 5e-3 * cost(md)
        # dist is the distance of the assignment it returns (emd_module.py:100-104).  (Not necessarily a permutation:
        # a handful of persons are still bidding when the 3000 rounds end, in the oracle as in the reference, and the
        # forced last round hands each its object whoever holds it, emd_cuda.cu:200.)
        ra_h = host(ra)
        assert ra_h.min() >= 0 and ra_h.max() < n
        want = ((a - np.take_along_axis(c, ra_h[..., None].astype(np.int64), 1)) ** 2).sum(-1)
        np.testing.assert_allclose(host(rd), want, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("kind", ["uniform", "chair gt + noise 0.03"])
def test_reference_emd_kernels_at_the_headline_size(mode, kind):
    """the headline size (16384 points, eps 0.004, 3000 rounds) against the reference's own emd_cuda.cu inside
    the suite, on a uniform pair and on a surface-shaped one (what the eval loop sees).  GetMax's race (emd_cuda.cu:188-191)
    makes the reference one member of an outcome SET here and the product another (the highest-bidder policy the oracle
    pins).  The reference is run TWICE on the same input to see its own spread -- measured (first run of this test): the
    two runs differ in 5 584 of 32 768 assignments but only 4.5e-7 in cost, because the same kernels on the same idle GPU
    replay nearly the same schedule, while the product's policy sits 1.2e-3 (relative) away on one cloud and 6.6e-5 on the
    other.  Two runs therefore UNDERSTATE the outcome set; its width was measured in round 2 by running the oracle under
    both extreme policies (profiles/r2_emd_schedule_sensitivity.txt: 1e-3 .. 2.3e-3).  Tolerance: the larger of three times
    the measured spread and 2.5e-3 of the cost (the smaller cases above use 5e-3)."""
    from mvp_benchmark_amd import metrics
    from mvp_benchmark_amd.synthetic import prediction_pair
    v, exact = mode
    b, n, eps, iters = 2, 16384, 0.004, 3000
    if kind == "uniform":
        a, c = dev(rand_clouds(7001, b, n, 3)), dev(rand_clouds(7002, b, n, 3))
    else:
        pred, gt = prediction_pair("chair", "0.03", torch.Generator().manual_seed(7003), b, n)
        a, c = pred.cuda(), gt.cuda()
    r1d, r1a, _ = ref.emd_forward(a, c, eps, iters, v)
    r2d, r2a, _ = ref.emd_forward(a, c, eps, iters, v)
    md, ma = metrics.emd()(a, c, eps, iters)
    cost = lambda d: np.sqrt(host(d).astype(np.float64)).mean(axis=1)            # per cloud
    c1, c2, cm = cost(r1d), cost(r2d), cost(md)
    differ = int((host(r1a) != host(r2a)).sum())
    # (Two agreeing runs do NOT make the reference deterministic here: on an idle GPU its racy GetMax can replay one
    # schedule twice -- seen in round 6, with a third of the assignments different from the product's pinned policy.  At this
    # size the comparison is always the cost-level one; the policy-free sizes are pinned exactly by EMD_CASES above.)
    spread = np.abs(c1 - c2)
    print("headline-size EMD vs the reference (%s, %s build): reference runs differ in %d of %d assignments, cost spread %s, "
          "product - reference mean %s (relative %s)" % (kind, v or "default", differ, b * n, spread, cm - (c1 + c2) / 2,
                                                         (cm - (c1 + c2) / 2) / cm))
    tol = np.maximum(3 * spread, 2.5e-3 * cm)
    assert (np.abs(cm - (c1 + c2) / 2) <= tol).all(), (cm, c1, c2)
    # and the product's dist is the distance of the assignment it returns
    mh = host(ma).astype(np.int64)
    assert mh.min() >= 0 and mh.max() < n
    want = ((host(a) - np.take_along_axis(host(c), mh[..., None], 1)) ** 2).sum(-1)
    np.testing.assert_allclose(host(md), want, rtol=1e-5, atol=1e-9)


def test_reference_emd_backward_kernel(oracle, mode):
    from mvp_benchmark_amd import metrics
    v, _ = mode
    b, n = 3, 1024
    a, c = rand_clouds(21, b, n, 3), rand_clouds(22, b, n, 3)
    _, ra, _ = ref.emd_forward(dev(a), dev(c), 0.005, 50, v)
    g = rand_clouds(23, b, n)
    gx1, gx2 = ref.emd_backward(dev(a), dev(c), dev(g), ra, v)
    np.testing.assert_array_equal(host(gx1), oracle.emd_backward(a, c, g, host(ra)))
    assert not host(gx2).any()                               # emd_module.py:76-81: gradxyz2 stays zero
    pa = dev(a).requires_grad_(True)
    d, asg = metrics.emd()(pa, dev(c), 0.005, 50)
    same_index(asg, ra)
    (d * dev(g)).sum().backward()
    np.testing.assert_array_equal(host(pa.grad), host(gx1))


def test_reference_kernels_behind_ref_gpu_reproduce_the_reference_wrappers_fixtures():
    """The two pins joined (parity seam b): tests/golden/ops_wrapper_golden.npz holds what the reference's own
    PYTHON WRAPPERS return (generated with their compiled extensions replaced by the oracle); here the same inputs go
    through oracle/ref_gpu.py -- the hand-made initial buffers (1e10 temp, zeroed idx, -1 assignments ...) in front of the
    reference's own KERNELS -- and must give the wrappers' outputs: every index identical, values within one ulp (the
    default-contraction build against fixtures computed in the canonical arithmetic).  A wrong initial value or argument
    order in ref_gpu.py would show here, independently of the oracle."""
    import os
    v = ""
    if not ref.available(v):
        pytest.skip("oracle/_ref is not built")
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ops_wrapper_golden.npz"))
    xyz, ctr, feat = dev(G["in/xyz"]), dev(G["in/ctr"]), dev(G["in/feat"])
    same_index(ref.ball_query(0.0, 0.2, 8, xyz, ctr, v), G["ball_query/full"])              # ball_query.py:35-38
    same_index(ref.ball_query(0.1, 0.25, 5, xyz, ctr, v), G["ball_query/ring"])
    idx, _ = ref.knn(5, xyz, ctr, v)                                                          # knn.py:55-72: (B, M, k) -> (B, k, M)
    same_index(idx.transpose(1, 2), G["knn/centres"])
    idx, _ = ref.knn(5, xyz, xyz, v)
    same_index(idx.transpose(1, 2), G["knn/self5"])
    d2, idx = ref.three_nn(xyz, ctr, v)                                                       # three_nn.py:31-45 (+ the wrapper's sqrt)
    same_index(idx, G["three/idx"])
    np.testing.assert_allclose(np.sqrt(host(d2)), G["three/dist"], rtol=2 * ULP, atol=0)
    np.testing.assert_allclose(host(ref.three_interpolate(dev(G["three/cfeat"]), dev(G["three/idx"]), dev(G["three/weight"]), v)),
                               G["three/out"], rtol=2 * ULP, atol=1e-7)       # (a three-term sum, contracted differently: one ulp of its O(1) terms)
    np.testing.assert_allclose(host(ref.three_interpolate_grad(dev(G["three/gy"]), dev(G["three/idx"]), dev(G["three/weight"]), 40, v)),
                               G["three/grad"], rtol=1e-5, atol=1e-6)
    same_index(ref.fps(xyz, 33, v), G["fps/idx"])                                             # furthest_point_sample.py:29-33
    same_index(ref.fps_with_dist(dev(G["fps/dmat"]), 20, v), G["fps/with_dist"])
    same_value(ref.gather_points(feat, dev(G["gather/idx"]), v), G["gather/out"], True)       # gather_points.py:27-31
    same_value(ref.grouping_operation(feat, dev(G["group/idx"]), v), G["group/out"], True)    # group_points.py:184-188
    np.testing.assert_allclose(host(ref.gather_points_grad(dev(G["gather/gy"]), dev(G["gather/idx"]), 300, v)), G["gather/grad"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(host(ref.grouping_operation_grad(dev(G["group/gy"]), dev(G["group/idx"]), 300, v)), G["group/grad"], rtol=1e-5, atol=1e-6)
    d1, d2c, i1, i2 = ref.chamfer_forward(dev(G["cd/a"]), dev(G["cd/b"]), v)                  # dist_chamfer_3D.py:30-50
    same_index(i1, G["cd/idx1"])
    same_index(i2, G["cd/idx2"])
    same_value(d1, G["cd/dist1"], False)
    same_value(d2c, G["cd/dist2"], False)
    p, q = dev(G["emd_in/p"]), dev(G["emd_in/q"])                                             # emd_module.py:49-65
    for tag, eps, iters in (("train", 0.005, 50), ("smoke", 0.05, 3000)):
        dist, ass, _ = ref.emd_forward(p, q, eps, iters, v)
        same_index(ass, G["emd_%s/assignment" % tag])
        same_value(dist, G["emd_%s/dist" % tag], False)
        gx1, _ = ref.emd_backward(p, q, dev(G["emd_%s/gd" % tag]), ass, v)
        np.testing.assert_allclose(host(gx1), G["emd_%s/grad_p" % tag], rtol=1e-5, atol=1e-6)


def test_hip_path_matches_committed_reference_kernel_outputs():
    """The same comparison against the COMMITTED outputs of the reference's kernels (tests/golden/ref_kernel_golden.npz,
    generated on an MI355X by tests/golden/make_ref_kernel_golden.py): needs no oracle/_ref on the box.  Indices identical
    (both builds of the reference agree on every one of them); values to the last place against the default build."""
    import os
    from conftest import GOLDEN
    from ref_kernel_cases import CASES, inputs
    from mvp_benchmark_amd import metrics, mm3d_pn2 as pn2
    blob = np.load(os.path.join(GOLDEN, "ref_kernel_golden.npz"))
    checked = 0
    for name, (kind, arg) in CASES.items():
        x = {k: dev(a) for k, a in inputs(name).items()}
        if kind == "fps":
            got = {"idx": pn2.furthest_point_sample(x["xyz"], arg["m"])}
        elif kind == "fps_dist":
            got = {"idx": pn2.furthest_point_sample_with_dist(x["dist"], arg["m"])}
        elif kind == "ball_query":
            got = {"idx": pn2.ball_query(arg["lo"], arg["hi"], arg["s"], x["xyz"], x["ctr"])}
        elif kind == "knn":
            got = {"idx": pn2.knn(arg["k"], x["xyz"], x["ctr"]).transpose(2, 1)}
        elif kind == "three_nn":
            d, i = pn2.three_nn(x["ctr"], x["xyz"])
            got = {"idx": i, "dist2": d * d}
        elif kind == "three_interpolate":
            got = {"out": pn2.three_interpolate(x["feat"], x["idx"], x["w"])}
        elif kind == "gather":
            got = {"out": pn2.gather_points(x["feat"], x["idx"])}
        elif kind == "group":
            got = {"out": pn2.grouping_operation(x["feat"], x["idx"])}
        elif kind == "chamfer":
            got = dict(zip(("dist1", "dist2", "idx1", "idx2"), metrics.cd()(x["a"], x["c"])))
        else:
            pa = x["a"].clone().requires_grad_(True)
            d, a = metrics.emd()(pa, x["c"], arg["eps"], arg["iters"])
            got = {"dist": d.detach(), "assignment": a}
            if arg.get("grad"):
                (d * x["g"]).sum().backward()
                got["gradxyz1"] = pa.grad
        for key, mine in got.items():
            want = blob[f"{name}/default/{key}"]
            if want.dtype.kind == "i":
                np.testing.assert_array_equal(host(mine), want, err_msg=f"{name}/{key}")
                np.testing.assert_array_equal(want, blob[f"{name}/_nofma/{key}"], err_msg=f"{name}/{key}: the two builds")
            else:
                rtol = 1e-6 if key == "dist2" and kind == "three_nn" else ULP    # (sqrt, then squared again here)
                np.testing.assert_allclose(host(mine), want, rtol=rtol, atol=0, err_msg=f"{name}/{key}")
            checked += 1
    assert checked >= 35
