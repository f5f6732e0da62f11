"""CPU tests that PIN THE ORACLE: golden vectors produced by the reference's
own CPU path (tests/golden/make_chamfer_golden.py), the reference's stated
contracts (unit_test.py:25-33, emd_module.py:100-104) and brute-force NumPy
restatements of each op's definition."""
import numpy as np
import pytest
from conftest import rand_clouds


# ------------------------------------------------------------------ chamfer
def test_chamfer_oracle_matches_reference_golden(oracle, chamfer_golden):
    """unit_test.py:25-33 contract: mean sq. diff < 1e-8 and indices EXACTLY
    equal to the reference's distChamfer, on every golden case."""
    for name, c in chamfer_golden.items():
        d1, d2, i1, i2 = oracle.chamfer_forward(c["a"], c["b"])
        err = np.mean((d1 - c["dist1"]) ** 2) + np.mean((d2 - c["dist2"]) ** 2)
        assert err < 1e-8, name
        np.testing.assert_array_equal(i1, c["idx1"], err_msg=name)
        np.testing.assert_array_equal(i2, c["idx2"], err_msg=name)
        # 1e-5 relative on the values themselves (north-star tolerance)
        np.testing.assert_allclose(d1, c["dist1"], rtol=1e-5, atol=1e-9, err_msg=name)
        np.testing.assert_allclose(d2, c["dist2"], rtol=1e-5, atol=1e-9, err_msg=name)


def test_chamfer_ties_lowest_index(oracle):
    a = np.zeros((1, 3, 3), np.float32)
    b = np.zeros((1, 5, 3), np.float32)
    b[0, 3] = 1.0
    d1, d2, i1, i2 = oracle.chamfer_forward(a, b)
    assert (i1 == 0).all() and (d1 == 0).all()
    np.testing.assert_array_equal(i2[0], [0, 0, 0, 0, 0])
    np.testing.assert_allclose(d2[0], [0, 0, 0, 3, 0])


def test_chamfer_backward_matches_autograd(oracle):
    import torch
    a = torch.tensor(rand_clouds(1, 2, 50, 3), requires_grad=True)
    b = torch.tensor(rand_clouds(2, 2, 70, 3), requires_grad=True)
    d1, d2, i1, i2 = oracle.chamfer_forward(a.detach().numpy(), b.detach().numpy())
    g1 = rand_clouds(3, 2, 50)
    g2 = rand_clouds(4, 2, 70)
    gx1, gx2 = oracle.chamfer_backward(a.detach().numpy(), b.detach().numpy(), g1, g2, i1, i2)
    bi1 = torch.tensor(i1).long()
    bi2 = torch.tensor(i2).long()
    t1 = ((a - torch.gather(b, 1, bi1[..., None].expand(-1, -1, 3))) ** 2).sum(-1)
    t2 = ((b - torch.gather(a, 1, bi2[..., None].expand(-1, -1, 3))) ** 2).sum(-1)
    ((t1 * torch.tensor(g1)).sum() + (t2 * torch.tensor(g2)).sum()).backward()
    np.testing.assert_allclose(gx1, a.grad.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(gx2, b.grad.numpy(), rtol=1e-4, atol=1e-6)


# ---------------------------------------------------------------------- emd
def test_emd_self_consistency_and_guards(oracle):
    """test_emd identity (emd_module.py:100-104): dist == |x1 - x2[assign]|^2;
    guards of emd_cuda.cu:236-249."""
    x1 = rand_clouds(0, 2, 1024, 3)
    x2 = rand_clouds(1, 2, 1024, 3)
    dist, ass = oracle.emd_forward(x1, x2, 0.005, 50)
    assert ass.min() >= 0 and ass.max() < 1024
    ref = ((x1 - np.take_along_axis(x2, ass[..., None].astype(np.int64), 1)) ** 2).sum(-1)
    np.testing.assert_allclose(dist, ref, rtol=1e-5, atol=1e-9)
    with pytest.raises(RuntimeError):
        oracle.emd_forward(x1[:, :1000], x2[:, :1000], 0.005, 50)


def test_emd_optimality_bound_vs_hungarian(oracle):
    """Auction theory: when the auction ends with every person assigned to a
    distinct object, total cost is within n*eps of the optimum."""
    from scipy.optimize import linear_sum_assignment
    n, eps = 1024, 0.002
    x1 = rand_clouds(5, 1, n, 3)
    x2 = rand_clouds(6, 1, n, 3)
    dist, ass, stats = oracle.emd_forward(x1, x2, eps, 20000, return_stats=True)
    assert stats[0, 0] < 20000, "auction should terminate before the forced round"
    assert len(np.unique(ass[0])) == n
    cost = np.sqrt(((x1[0][:, None] - x2[0][None]) ** 2).sum(-1))
    r, c = linear_sum_assignment(cost)
    opt = cost[r, c].sum()
    got = np.sqrt(dist[0]).sum()
    assert opt - 1e-3 <= got <= opt + n * eps + 1e-3


def test_emd_forced_last_round_assigns_everyone(oracle):
    x1 = rand_clouds(7, 1, 1024, 3)
    x2 = rand_clouds(8, 1, 1024, 3)
    dist, ass = oracle.emd_forward(x1, x2, 0.005, 1)
    assert (ass >= 0).all()          # one round, forced: everyone bid-assigned
    assert len(np.unique(ass)) < 1024  # and therefore not injective


def test_emd_backward(oracle):
    x1 = rand_clouds(9, 1, 1024, 3)
    x2 = rand_clouds(10, 1, 1024, 3)
    dist, ass = oracle.emd_forward(x1, x2, 0.005, 50)
    g = rand_clouds(11, 1, 1024)
    gx = oracle.emd_backward(x1, x2, g, ass)
    ref = 2 * g[..., None] * (x1 - np.take_along_axis(x2, ass[..., None].astype(np.int64), 1))
    np.testing.assert_allclose(gx, ref, rtol=1e-5, atol=1e-7)


# ---------------------------------------------------------------------- fps
def _fps_numpy(x, m):
    """Textbook greedy max-min FPS; ties (never hit on random data) -> lowest."""
    n = x.shape[0]
    temp = np.full(n, 1e10, np.float32)
    idx = [0]
    for _ in range(1, m):
        d = ((x - x[idx[-1]]) ** 2).sum(-1).astype(np.float32)
        temp = np.minimum(temp, d)
        idx.append(int(np.argmax(temp)))
    return np.array(idx)


@pytest.mark.parametrize("n,m", [(64, 16), (100, 100), (777, 50), (2048, 128), (3072, 96)])
def test_fps_is_greedy_max_min(oracle, n, m):
    x = rand_clouds(n, 2, n, 3)
    idx = oracle.furthest_point_sample(x, m)
    assert idx.shape == (2, m) and (idx[:, 0] == 0).all()
    for b in range(2):
        temp = np.full(n, np.inf)
        for j in range(1, m):
            d = ((x[b].astype(np.float64) - x[b, idx[b, j - 1]]) ** 2).sum(-1)
            temp = np.minimum(temp, d)
            assert temp[idx[b, j]] >= temp.max() * (1 - 1e-5)
        if m <= n:
            assert len(np.unique(idx[b])) == m


def test_fps_block_size_rule(oracle):
    # opt_n_threads (furthest_point_sample_cuda.cu:11-15)
    for n, bs in [(1, 1), (2, 2), (3, 2), (64, 64), (100, 64), (1000, 512),
                  (1024, 1024), (2048, 1024), (3072, 1024), (16384, 1024)]:
        assert oracle.fps_block_size(n) == bs, n
    # the double log ratio truncates below the exact power for some n
    assert oracle.fps_block_size(8) in (4, 8)


def test_fps_tie_rule_bit_reversed_slot(oracle):
    """Equal maxima: the LDS tree (furthest_point_sample_cuda.cu:17-23) keeps
    the slot with the smallest bit-reversed thread id.  8 points, first at the
    origin, points 1, 2 and 4 on the unit sphere (equal distance), rest near:
    slots {1,2,4} tie -> slot 4 (bit-reversed 001) wins."""
    x = np.zeros((1, 8, 3), np.float32)
    x[0, 1] = [1, 0, 0]
    x[0, 2] = [0, 1, 0]
    x[0, 4] = [0, 0, 1]
    x[0, 3] = [0.1, 0, 0]
    x[0, 5] = [0, 0.1, 0]
    x[0, 6] = [0, 0, 0.1]
    x[0, 7] = [0.1, 0.1, 0]
    if oracle.fps_block_size(8) != 8:
        pytest.skip("libm rounds log(8)/log(2) below 3 here")
    idx = oracle.furthest_point_sample(x, 2)
    assert idx[0, 1] == 4


def test_fps_with_dist_equals_fps_on_same_metric(oracle):
    x = rand_clouds(3, 2, 300, 3)
    # distance matrix built with the oracle's own chain so both paths see the
    # same floats
    dx = x[:, None, :, 0] - x[:, :, None, 0]
    dy = x[:, None, :, 1] - x[:, :, None, 1]
    dz = x[:, None, :, 2] - x[:, :, None, 2]
    # emulate fmaf(dz,dz, fmaf(dy,dy, dx*dx)) in float64 then round: exact
    # enough to preserve the arg-max on random data
    dist = (dx.astype(np.float64) ** 2 + dy.astype(np.float64) ** 2 + dz.astype(np.float64) ** 2).astype(np.float32)
    a = oracle.furthest_point_sample(x, 40)
    b = oracle.furthest_point_sample_with_dist(dist, 40)
    np.testing.assert_array_equal(a, b)


# ------------------------------------------------------- ball_query/knn/3nn
def _sqd(c, p):
    return ((c[:, :, None, :].astype(np.float64) - p[:, None, :, :].astype(np.float64)) ** 2).sum(-1)


def test_ball_query_definition(oracle):
    xyz = rand_clouds(0, 2, 500, 3)
    ctr = xyz[:, :40].copy()
    ctr[:, 5] += 10.0  # a centre with no neighbour at all
    r0, r1, S = 0.05, 0.2, 8
    idx = oracle.ball_query(r0, r1, S, xyz, ctr)
    d2 = _sqd(ctr, xyz)
    for b in range(2):
        for p in range(40):
            ok = np.where((d2[b, p] == 0) | ((d2[b, p] >= np.float32(r0) ** 2) & (d2[b, p] < np.float32(r1) ** 2)))[0]
            want = np.zeros(S, np.int32)
            if len(ok):
                want[:] = ok[0]
                want[:min(S, len(ok))] = ok[:S]
            np.testing.assert_array_equal(idx[b, p], want)
    assert (idx[:, 5] == 0).all()


@pytest.mark.parametrize("k", [1, 5, 16, 33])
def test_knn_matches_argsort(oracle, k):
    xyz = rand_clouds(1, 2, 400, 3)
    ctr = rand_clouds(2, 2, 30, 3)
    idx, dist2 = oracle.knn(k, xyz, ctr, return_dist=True)
    assert idx.shape == (2, k, 30)
    d2 = _sqd(ctr, xyz)
    order = np.argsort(d2, axis=-1, kind="stable")[..., :k]
    np.testing.assert_array_equal(idx.transpose(0, 2, 1), order)
    np.testing.assert_allclose(dist2, np.take_along_axis(d2, order, -1), rtol=1e-5)
    # default centre = xyz, transposed inputs
    a = oracle.knn(3, xyz)
    b = oracle.knn(3, xyz.transpose(0, 2, 1), None, True)
    np.testing.assert_array_equal(a, b)
    assert (a[:, 0] == np.arange(400)).all()


def test_three_nn_matches_argsort_and_small_m(oracle):
    tgt = rand_clouds(3, 2, 200, 3)
    src = rand_clouds(4, 2, 90, 3)
    dist, idx = oracle.three_nn(tgt, src)
    d2 = _sqd(tgt, src)
    order = np.argsort(d2, axis=-1, kind="stable")[..., :3]
    np.testing.assert_array_equal(idx, order)
    np.testing.assert_allclose(dist, np.sqrt(np.take_along_axis(d2, order, -1)), rtol=1e-5)
    # fewer than 3 sources: unused slots are (float)1e40 = inf, index 0
    dist, idx = oracle.three_nn(tgt, src[:, :2])
    assert np.isinf(dist[..., 2]).all() and (idx[..., 2] == 0).all()


# ------------------------------------------------- gather/group/interpolate
def test_gather_group_interpolate_definitions(oracle):
    rng = np.random.default_rng(0)
    f = rand_clouds(5, 2, 7, 50)
    gi = rng.integers(0, 50, (2, 33)).astype(np.int32)
    out = oracle.gather_points(f, gi)
    np.testing.assert_array_equal(out, np.take_along_axis(f, gi[:, None, :].astype(np.int64).repeat(7, 1), 2))
    go = rand_clouds(6, 2, 7, 33)
    gp = oracle.gather_points_grad(go, gi, 50)
    want = np.zeros((2, 7, 50), np.float64)
    for b in range(2):
        for p in range(33):
            want[b, :, gi[b, p]] += go[b, :, p]
    np.testing.assert_allclose(gp, want, rtol=1e-5, atol=1e-6)

    qi = rng.integers(0, 50, (2, 11, 4)).astype(np.int32)
    out = oracle.grouping_operation(f, qi)
    assert out.shape == (2, 7, 11, 4)
    for b in range(2):
        np.testing.assert_array_equal(out[b], f[b][:, qi[b]])
    go = rand_clouds(7, 2, 7, 11, 4)
    gp = oracle.grouping_operation_grad(go, qi, 50)
    want = np.zeros((2, 7, 50), np.float64)
    for b in range(2):
        for p in range(11):
            for s in range(4):
                want[b, :, qi[b, p, s]] += go[b, :, p, s]
    np.testing.assert_allclose(gp, want, rtol=1e-5, atol=1e-6)

    ti = rng.integers(0, 50, (2, 21, 3)).astype(np.int32)
    w = rand_clouds(8, 2, 21, 3)
    out = oracle.three_interpolate(f, ti, w)
    want = np.zeros((2, 7, 21), np.float64)
    for b in range(2):
        for j in range(3):
            want[b] += w[b, :, j][None] * f[b][:, ti[b, :, j]]
    np.testing.assert_allclose(out, want, rtol=1e-5, atol=1e-6)
    go = rand_clouds(9, 2, 7, 21)
    gp = oracle.three_interpolate_grad(go, ti, w, 50)
    want = np.zeros((2, 7, 50), np.float64)
    for b in range(2):
        for p in range(21):
            for j in range(3):
                want[b, :, ti[b, p, j]] += go[b, :, p] * w[b, p, j]
    np.testing.assert_allclose(gp, want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("n,eps,iters", [(1024, 0.004, 3000), (2048, 0.004, 3000), (1024, 0.005, 50)])
def test_emd_sensitivity_to_the_getmax_schedule(n, eps, iters):
    """GetMax (emd_cuda.cu:181-194) lets the LAST writer among the bidders within
    1e-6 of an object's maximal increment win -- a data race in the reference.
    The oracle (and the HIP kernel) resolve it to the highest qualifying bidder,
    i.e. the reference executed sequentially.  Here the other extreme -- the
    LOWEST qualifying bidder -- is run beside it.  Measured (this test,
    tools/emd_schedule_sensitivity.py, profiles/r2_emd_schedule_sensitivity.txt):
    most clouds never see two bidders inside one band and are bit-identical under
    both schedules; a cloud that does diverges in 10-20 % of its assignments and
    its mean(sqrt(dist)) moves by a few 1e-4 relative.  So the reference's own
    run-to-run spread is ~1e-3, two orders above the north star's 1e-5 band: that
    band can only be (and is) met per schedule -- HIP == oracle bit for bit on the
    pinned schedule -- not across schedules."""
    import oracle
    rng = np.random.default_rng(n + iters)
    x1 = rng.random((4, n, 3), dtype=np.float32)
    x2 = rng.random((4, n, 3), dtype=np.float32)
    d_hi, a_hi, _, _ = oracle.emd_forward_ex(x1, x2, eps, iters, getmax_lowest=False)
    d_lo, a_lo, _, _ = oracle.emd_forward_ex(x1, x2, eps, iters, getmax_lowest=True)
    m_hi, m_lo = np.sqrt(d_hi).mean(1), np.sqrt(d_lo).mean(1)
    same = (a_hi == a_lo).all(1)
    assert same.sum() >= 2                                        # most clouds: no in-band meeting at all
    np.testing.assert_array_equal(d_hi[same], d_lo[same])
    np.testing.assert_allclose(m_lo, m_hi, rtol=2e-3)             # the others: the reference's own spread
    # both schedules are valid auctions: dist is the squared length of the matched pair
    for d, a in ((d_hi, a_hi), (d_lo, a_lo)):
        m = np.take_along_axis(x2, a[..., None].astype(np.int64), 1)
        np.testing.assert_allclose(d, ((x1 - m) ** 2).sum(-1), rtol=1e-5, atol=1e-9)
    d0, a0 = oracle.emd_forward(x1, x2, eps, iters)               # the default entry point is the "highest" schedule
    np.testing.assert_array_equal(a0, a_hi)
    np.testing.assert_array_equal(d0, d_hi)


def test_emd_getmax_schedule_matters_when_increments_tie():
    """The two schedules really are different programs: duplicated persons bid
    identical increments on the same objects, and the lowest / highest bidder
    wins respectively -- the assignments differ, the matched cost does not."""
    import oracle
    rng = np.random.default_rng(3)
    base1 = rng.random((1, 512, 3), dtype=np.float32)
    x1 = np.tile(base1, (1, 2, 1))                               # every person twice
    x2 = rng.random((1, 1024, 3), dtype=np.float32)
    d_hi, a_hi, _, _ = oracle.emd_forward_ex(x1, x2, 0.005, 200, getmax_lowest=False)
    d_lo, a_lo, _, _ = oracle.emd_forward_ex(x1, x2, 0.005, 200, getmax_lowest=True)
    assert (a_hi != a_lo).any()
    assert np.sqrt(d_lo).mean() == pytest.approx(np.sqrt(d_hi).mean(), rel=2e-2)


# ------------------------------------------- the reference's own kernels, run on an MI355X
def _oracle_outputs(oracle, kind, arg, x):
    if kind == "fps":
        return {"idx": oracle.furthest_point_sample(x["xyz"], arg["m"])}
    if kind == "fps_dist":
        return {"idx": oracle.furthest_point_sample_with_dist(x["dist"], arg["m"])}
    if kind == "ball_query":
        return {"idx": oracle.ball_query(arg["lo"], arg["hi"], arg["s"], x["xyz"], x["ctr"])}
    if kind == "knn":
        i, d = oracle.knn(arg["k"], x["xyz"], x["ctr"], return_dist=True)
        return {"idx": i.transpose(0, 2, 1), "dist2": d}
    if kind == "three_nn":
        d, i = oracle.three_nn(x["ctr"], x["xyz"])
        return {"idx": i, "dist": d}
    if kind == "three_interpolate":
        return {"out": oracle.three_interpolate(x["feat"], x["idx"], x["w"])}
    if kind == "gather":
        return {"out": oracle.gather_points(x["feat"], x["idx"])}
    if kind == "group":
        return {"out": oracle.grouping_operation(x["feat"], x["idx"])}
    if kind == "chamfer":
        return dict(zip(("dist1", "dist2", "idx1", "idx2"), oracle.chamfer_forward(x["a"], x["c"])))
    if kind == "emd":
        d, a = oracle.emd_forward(x["a"], x["c"], arg["eps"], arg["iters"])
        res = {"dist": d, "assignment": a}
        if arg.get("grad"):
            res["gradxyz1"] = oracle.emd_backward(x["a"], x["c"], x["g"], a)
        return res
    raise KeyError(kind)


@pytest.mark.parametrize("build", ["_nofma", "default"])
def test_oracle_matches_reference_kernel_outputs(oracle, build):
    """tests/golden/ref_kernel_golden.npz holds what the REFERENCE'S OWN KERNELS produced on an MI355X
    (oracle/build_ref_gpu.sh + tests/golden/make_ref_kernel_golden.py), for 23 seeded cases of every op.
    Built without contraction ("_nofma") they must equal the oracle in its no-contraction mode BIT FOR BIT, values
    included; built with hipcc's default contraction every index must equal the oracle's (canonical mode) and every
    value agree to the last place (2.5e-7; north_star 1e-5)."""
    import os
    from conftest import GOLDEN
    from ref_kernel_cases import CASES, inputs
    blob = np.load(os.path.join(GOLDEN, "ref_kernel_golden.npz"))
    exact = build == "_nofma"
    oracle.set_contraction(not exact)
    try:
        checked = 0
        for name, (kind, arg) in CASES.items():
            got = _oracle_outputs(oracle, kind, arg, inputs(name))
            for key, mine in got.items():
                fkey = "dist2" if (kind == "three_nn" and key == "dist") else key
                want = blob[f"{name}/{build}/{fkey}"]
                if kind == "three_nn" and key == "dist":
                    want = np.sqrt(want)              # three_nn.py:38, the wrapper's host-side sqrt
                if want.dtype.kind == "i" or exact:
                    np.testing.assert_array_equal(mine, want.astype(mine.dtype), err_msg=f"{name}/{key}")
                else:
                    np.testing.assert_allclose(mine, want, rtol=2.5e-7, atol=0, err_msg=f"{name}/{key}")
                checked += 1
        assert checked >= 40
    finally:
        oracle.set_contraction(True)
