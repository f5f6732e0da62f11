"""GPU parity tests (run with -m gpu on an MI355X): every operator, called
through the reference-shaped Python API -> ctypes -> C ABI -> HIP kernels, is
compared with the CPU oracle on identical seeded inputs.  Index outputs and
forward values are BIT-EXACT (the kernels and the oracle share one canonical
arithmetic); gradients that use float atomics are compared at 1e-5."""
import math

import numpy as np
import pytest
import torch
from conftest import rand_clouds

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def test_extension_is_loaded_not_a_fallback():
    from mvp_benchmark_amd import _lib
    lib = _lib.load()
    assert lib.mvp_abi_version() == _lib.ABI_VERSION
    assert "libmvpops.so" in open("/proc/self/maps").read()
    assert torch.cuda.is_available()


# ------------------------------------------------------------------ chamfer
def _cd_check(oracle, a, b):
    from mvp_benchmark_amd.metrics import cd
    d1, d2, i1, i2 = cd()(dev(a), dev(b))
    o1, o2, j1, j2 = oracle.chamfer_forward(a, b)
    np.testing.assert_array_equal(i1.cpu().numpy(), j1)
    np.testing.assert_array_equal(i2.cpu().numpy(), j2)
    np.testing.assert_array_equal(d1.cpu().numpy(), o1)
    np.testing.assert_array_equal(d2.cpu().numpy(), o2)
    assert d1.dtype == torch.float32 and i1.dtype == torch.int32
    return d1, d2, i1, i2


def test_chamfer_golden_vectors(oracle, chamfer_golden):
    """HIP path vs the vectors produced by the reference's distChamfer under
    its own contract (unit_test.py:25-33): indices exactly equal, values
    within 1e-5 relative."""
    for name, c in chamfer_golden.items():
        d1, d2, i1, i2 = _cd_check(oracle, c["a"], c["b"])
        np.testing.assert_array_equal(i1.cpu().numpy(), c["idx1"], err_msg=name)
        np.testing.assert_array_equal(i2.cpu().numpy(), c["idx2"], err_msg=name)
        np.testing.assert_allclose(d1.cpu().numpy(), c["dist1"], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(d2.cpu().numpy(), c["dist2"], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("b,n,m", [(1, 1, 1), (2, 17, 1000), (3, 1000, 17),
                                   (2, 1024, 1024), (1, 1025, 2049),
                                   (8, 2048, 2048), (32, 2000, 1000),
                                   (2, 16384, 2048), (70, 3072, 2048)])
def test_chamfer_matches_oracle(oracle, b, n, m):
    _cd_check(oracle, rand_clouds(n, b, n, 3), rand_clouds(m + 1, b, m, 3))


@pytest.mark.parametrize("b,n,m", [(2, 4096, 4096), (3, 8192, 2500), (2, 3000, 6000), (1, 16384, 16384)])
def test_chamfer_sorted_variant_is_bit_identical(oracle, b, n, m):
    """Clouds of >= 2048 points per side and >= 2^24 pairs take the Morton-sorted, tile-skipping
    kernel: distances AND indices equal the exhaustive kernel's / the oracle's."""
    from mvp_benchmark_amd import _lib
    a, c = rand_clouds(n + 3, b, n, 3), rand_clouds(m + 5, b, m, 3)
    ta, tc = dev(a), dev(c)
    outs = []
    for name in ("mvp_chamfer_forward", "mvp_chamfer_forward_sorted"):
        d1, d2 = torch.zeros(b, n, device=DEV), torch.zeros(b, m, device=DEV)
        i1 = torch.zeros(b, n, dtype=torch.int32, device=DEV)
        i2 = torch.zeros(b, m, dtype=torch.int32, device=DEV)
        if name.endswith("sorted"):
            nbytes = _lib.chamfer_scratch_bytes(b, n, m)
            scratch = torch.full((nbytes,), 0xAB, dtype=torch.uint8, device=DEV)   # contents are irrelevant
            _lib.call(name, ta.device, b, n, m, ta, tc, d1, d2, i1, i2, scratch, nbytes)
        else:
            _lib.call(name, ta.device, b, n, m, ta, tc, d1, d2, i1, i2)
        outs.append((d1, d2, i1, i2))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    if True:   # also at the headline size (1, 16384, 16384): ~6 s of one core
        o1, o2, j1, j2 = oracle.chamfer_forward(a, c)
        np.testing.assert_array_equal(outs[1][0].cpu().numpy(), o1)
        np.testing.assert_array_equal(outs[1][2].cpu().numpy(), j1)
        np.testing.assert_array_equal(outs[1][3].cpu().numpy(), j2)


def test_chamfer_sorted_variant_ties_and_clusters(oracle):
    """Lattice points (exact distance ties -> lowest original index), duplicated
    points and a tight cluster against a spread cloud (no tile can be skipped)."""
    from mvp_benchmark_amd.metrics import cd
    g = np.stack(np.meshgrid(*[np.arange(16)] * 3, indexing="ij"), -1).reshape(1, -1, 3).astype(np.float32) / 16
    lattice = np.ascontiguousarray(g[:, np.random.default_rng(0).permutation(4096)])
    shifted = np.ascontiguousarray((g + np.float32(1 / 32))[:, np.random.default_rng(1).permutation(4096)])
    dup = np.tile(rand_clouds(2, 1, 1024, 3), (1, 4, 1))
    tight = (0.5 + 0.001 * rand_clouds(3, 1, 4096, 3)).astype(np.float32)
    for a, c in [(lattice, shifted), (dup, rand_clouds(4, 1, 4096, 3)), (tight, rand_clouds(5, 1, 4096, 3)), (lattice, lattice)]:
        d1, d2, i1, i2 = cd()(dev(a), dev(c))
        o1, o2, j1, j2 = oracle.chamfer_forward(a, c)
        np.testing.assert_array_equal(d1.cpu().numpy(), o1)
        np.testing.assert_array_equal(d2.cpu().numpy(), o2)
        np.testing.assert_array_equal(i1.cpu().numpy(), j1)
        np.testing.assert_array_equal(i2.cpu().numpy(), j2)


def test_chamfer_exact_ties(oracle):
    g = np.stack(np.meshgrid(*[np.arange(6) / 8.0] * 3, indexing="ij"), -1).reshape(1, -1, 3).astype(np.float32)
    a = np.concatenate([g, g], 1)
    b = np.concatenate([g + 1 / 16.0, g + 1 / 16.0, g + 1 / 16.0], 1).astype(np.float32)
    _cd_check(oracle, a, b)


@pytest.mark.parametrize("bsz,n,m", [(4, 700, 300), (3, 2048, 2048), (2, 1, 8191), (2, 6000, 5000), (2, 4097, 4096)])
def test_chamfer_backward(oracle, bsz, n, m):
    """n + m <= 8192: gradients accumulated in LDS per cloud; above: global
    float atomics.  The (1, 8191) case scatters every point of one side onto a
    single point of the other."""
    from mvp_benchmark_amd.metrics import cd
    a, b = rand_clouds(1, bsz, n, 3), rand_clouds(2, bsz, m, 3)
    ta, tb = dev(a).requires_grad_(), dev(b).requires_grad_()
    d1, d2, i1, i2 = cd()(ta, tb)
    g1, g2 = rand_clouds(3, bsz, n), rand_clouds(4, bsz, m)
    (d1 * dev(g1)).sum().add((d2 * dev(g2)).sum()).backward()
    gx1, gx2 = oracle.chamfer_backward(a, b, g1, g2, i1.cpu().numpy(), i2.cpu().numpy())
    np.testing.assert_allclose(ta.grad.cpu().numpy(), gx1, rtol=1e-5, atol=1e-6 * max(1, m // 64))
    np.testing.assert_allclose(tb.grad.cpu().numpy(), gx2, rtol=1e-5, atol=1e-6 * max(1, n // 64))


def test_chamfer_full_size_properties():
    """BASELINE headline shape (64, 16384) x (64, 16384): size-independent
    checks -- symmetry of the two directions, idx consistency, self-distance."""
    from mvp_benchmark_amd.metrics import cd
    g = torch.Generator().manual_seed(0)
    a = torch.rand(64, 16384, 3, generator=g).to(DEV)
    b = torch.rand(64, 16384, 3, generator=g).to(DEV)
    d1, d2, i1, i2 = cd()(a, b)
    e2, e1, k2, k1 = cd()(b, a)          # swapped roles must swap outputs
    assert torch.equal(d1, e1) and torch.equal(d2, e2)
    assert torch.equal(i1, k1) and torch.equal(i2, k2)
    nb = torch.gather(b, 1, i1.long()[..., None].expand(-1, -1, 3))
    ref = ((nb - a) ** 2).sum(-1)
    assert torch.allclose(d1, ref, rtol=1e-5, atol=1e-9)
    # dist1[j] <= distance to 64 random candidates
    probe = torch.randint(0, 16384, (64,), device=DEV)
    dp = ((a[:, :, None, :] - b[:, probe][:, None, :, :]) ** 2).sum(-1)
    assert (d1[..., None] <= dp * (1 + 1e-6)).all()
    z1, z2, s1, s2 = cd()(a, a)
    assert (z1 == 0).all() and (z2 == 0).all()
    assert torch.equal(s1, torch.arange(16384, device=DEV, dtype=torch.int32).expand(64, -1))


# ---------------------------------------------------------------------- emd
@pytest.mark.parametrize("b,n,eps,iters", [(2, 1024, 0.005, 50), (3, 2048, 0.004, 3000),
                                           (1, 1024, 0.002, 10000), (2, 3072, 0.005, 1),
                                           (2, 1024, 0.05, 200), (1, 8192, 0.004, 3000)])
def test_emd_matches_oracle_bit_exact(oracle, b, n, eps, iters):
    from mvp_benchmark_amd.metrics import emd
    x1, x2 = rand_clouds(n, b, n, 3), rand_clouds(n + 1, b, n, 3)
    dist, ass = emd()(dev(x1), dev(x2), eps, iters)
    od, oa = oracle.emd_forward(x1, x2, eps, iters)
    np.testing.assert_array_equal(ass.cpu().numpy(), oa)
    np.testing.assert_array_equal(dist.cpu().numpy(), od)


@pytest.mark.parametrize("kind", ["uniform", "sheet"])
@pytest.mark.parametrize("b,n,eps,iters", [(1, 17408, 0.01, 40), (2, 33792, 0.02, 5), (1, 66560, 0.05, 2)])
def test_emd_above_16384_points_matches_oracle(oracle, emd_split, b, n, eps, iters, kind):
    """More than 16384 points: a leaf of the index holds 32 / 64 / 128 slots (csrc/emd_index.h keeps the leaves at <= 1024:
    544 / 528 / 520 of them here, the last node of the third case half empty), a visited leaf is several 16-slot chunks, the
    re-scan of a winner's leaf loops, the rounds stay plain (gathered bids need <= 16384).  Both launch sequences; a flat
    sheet as the second cloud kind (every box degenerate in z)."""
    from mvp_benchmark_amd.metrics import emd
    x1, x2 = rand_clouds(n + 7, b, n, 3), rand_clouds(n + 8, b, n, 3)
    if kind == "sheet":
        x1[..., 2] = 0.5
        x2[..., 2] = 0.5
    od, oa = oracle.emd_forward(x1, x2, eps, iters)
    for split in (0, 5):
        emd_split(split)
        dist, ass = emd()(dev(x1), dev(x2), eps, iters)
        np.testing.assert_array_equal(ass.cpu().numpy(), oa)
        np.testing.assert_array_equal(dist.cpu().numpy(), od)


def _hilbert_keys(pts, lo, scale, bits=9):
    """NumPy restatement of csrc/emd_index.h: emd_hilbert_key (Skilling's transpose form, three axes)."""
    top = (1 << bits) - 1
    c = np.clip(((pts - lo) * scale).astype(np.int32), 0, top).astype(np.uint32)
    x, y, z = c[:, 0].copy(), c[:, 1].copy(), c[:, 2].copy()
    q = np.uint32(1 << (bits - 1))
    while q > 1:
        p_ = np.uint32(q - 1)
        x = np.where(x & q, x ^ p_, x)
        t = (x ^ y) & p_
        x, y = np.where(y & q, x ^ p_, x ^ t), np.where(y & q, y, y ^ t)
        t = (x ^ z) & p_
        x, z = np.where(z & q, x ^ p_, x ^ t), np.where(z & q, z, z ^ t)
        q = np.uint32(q >> 1)
    y = y ^ x
    z = z ^ y
    t = np.zeros_like(x)
    q = np.uint32(1 << (bits - 1))
    while q > 1:
        t = np.where(z & q, t ^ np.uint32(q - 1), t)
        q = np.uint32(q >> 1)
    x, y, z = x ^ t, y ^ t, z ^ t
    h = np.zeros_like(x)
    for b_ in range(bits - 1, -1, -1):
        h = (h << np.uint32(3)) | (((x >> np.uint32(b_)) & np.uint32(1)) << np.uint32(2)) | (((y >> np.uint32(b_)) & np.uint32(1)) << np.uint32(1)) | ((z >> np.uint32(b_)) & np.uint32(1))
    return h


@pytest.mark.parametrize("n,kind", [(1024, "uniform"), (4096, "chair"), (16384, "uniform"), (20480, "uniform")])
def test_emd_index_is_a_hilbert_sort_of_the_objects(n, kind):
    """The index build of round 6 (csrc/emd_index.h), read back from the call's scratch: the sorted objects are a permutation
    of xyz2 (perm), they are in non-decreasing order of the 27-bit Hilbert key of the cube that bounds both clouds (the
    three-pass radix sort), and every person's home chunk is the chunk whose key range holds the person's own key."""
    from mvp_benchmark_amd import _lib
    from mvp_benchmark_amd.synthetic import prediction_pair
    b = 2
    if kind == "uniform":
        x1, x2 = rand_clouds(n + 31, b, n, 3), rand_clouds(n + 32, b, n, 3)
    else:
        p, g_ = prediction_pair("chair", "0.03", torch.Generator().manual_seed(n), b, n)
        x1, x2 = p.numpy(), g_.numpy()
    nbytes = _lib.emd_scratch_bytes(b, n)
    scratch = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    dist = torch.zeros(b, n, device=DEV)
    ass = torch.zeros(b, n, dtype=torch.int32, device=DEV)
    _lib.call("mvp_emd_forward", DEV, b, n, dev(x1), dev(x2), dist, ass, 0.004, 3, scratch, nbytes)
    torch.cuda.synchronize()
    per_cloud = (nbytes - b * (768 + 2 * (2 * 256 + 8) * 8 + 96 + 16)) // b
    lshift = 4
    while (n >> lshift) > 1024:
        lshift += 1
    for c in range(b):
        base = scratch[c * per_cloud: (c + 1) * per_cloud]
        obj = base[: n * 16].view(torch.float32).view(n, 4).cpu().numpy()
        person = base[n * 32: n * 64].view(torch.float32).view(n, 8).cpu().numpy()
        perm = base[n * 64: n * 68].view(torch.int32).cpu().numpy()
        assert sorted(perm.tolist()) == list(range(n))
        np.testing.assert_array_equal(obj[:, :3], x2[c][perm])
        both = np.concatenate([x1[c], x2[c]], 0)
        lo = both.min(0)
        ext = np.float32((both.max(0) - lo).max())
        scale = np.float32(512.0) / ext
        keys = _hilbert_keys(obj[:, :3], lo, scale)
        assert (np.diff(keys.astype(np.int64)) >= 0).all()
        # home chunk of every person: the last leaf whose first key is <= the person's key (leaf 0 if none), as a chunk
        leaf_first = keys[:: 1 << lshift].astype(np.int64)
        pk = _hilbert_keys(x1[c], lo, scale).astype(np.int64)
        want = np.maximum(np.searchsorted(leaf_first, pk, side="right") - 1, 0) << (lshift - 4)
        np.testing.assert_array_equal(person[:, :3], x1[c])
        np.testing.assert_array_equal(person[:, 3].view(np.int32), want.astype(np.int32))


@pytest.fixture
def cluster_width():
    """Pins how many workgroups own one cloud (mvp_emd_configure; automatic again afterwards)."""
    from mvp_benchmark_amd import _lib
    yield lambda w: _lib.emd_configure(cluster=w)
    _lib.emd_configure(cluster=0)


@pytest.fixture
def emd_split():
    """Selects whether the tail rounds run in the lean second kernel (mvp_emd_configure(split));
    0: never; 1: yes, fixed cluster widths; 2: yes, widths dealt out again at round 300 when the batch
    allows it); default restored afterwards."""
    from mvp_benchmark_amd import _lib
    yield lambda split: _lib.emd_configure(split=split)
    _lib.emd_configure(split=_lib.EMD_DEFAULT_SPLIT)


@pytest.mark.parametrize("width", [1, 2, 4, 8])
@pytest.mark.parametrize("kind", ["uniform", "clustered", "duplicates"])
def test_emd_every_cluster_width_matches_oracle(oracle, cluster_width, width, kind):
    """Same bits whether one, two, four or eight workgroups share a cloud.  The
    clustered case keeps thousands of bidders on a handful of objects (bid
    increments within the reference's 1e-6 GetMax band, more bidders per
    workgroup than the LDS bid cache holds); duplicates force value ties."""
    from mvp_benchmark_amd.metrics import emd
    cluster_width(width)
    if kind == "uniform":
        x1, x2, eps, iters = rand_clouds(11, 3, 4096, 3), rand_clouds(12, 3, 4096, 3), 0.004, 3000
    elif kind == "clustered":
        x1 = (0.5 + 0.01 * rand_clouds(13, 2, 8192, 3)).astype(np.float32)
        x2, eps, iters = rand_clouds(14, 2, 8192, 3), 0.004, 40
    else:
        x1 = np.tile(rand_clouds(15, 1, 512, 3), (1, 4, 1))
        x2, eps, iters = np.tile(rand_clouds(16, 1, 256, 3), (1, 8, 1)), 0.005, 300
    dist, ass = emd()(dev(x1), dev(x2), eps, iters)
    od, oa = oracle.emd_forward(x1, x2, eps, iters)
    np.testing.assert_array_equal(ass.cpu().numpy(), oa)
    np.testing.assert_array_equal(dist.cpu().numpy(), od)


def _handover_round(oracle, x1, x2, eps, width):
    """First round the lean kernel runs for every cloud of the batch (oracle trace of the
    unassigned counts): the hand-over needs <= 384 unassigned persons in total and <= 96 per
    workgroup; with the persons spread evenly over `width` lists that is the first round that
    starts with <= min(384, 96 * width) of them (an upper bound of the real hand-over round is
    enough for the tests below: they only need to straddle it)."""
    trace = oracle.emd_forward_ex(x1, x2, eps, 3000)[3]
    cap = min(384, 96 * width)
    return max(int(np.argmax(row <= cap)) for row in trace)


@pytest.mark.parametrize("split", [0, 1])
@pytest.mark.parametrize("width", [1, 2, 4, 8])
@pytest.mark.parametrize("kind", ["uniform", "duplicates", "few_rounds_left", "last_round_forced", "person_blob",
                                  "object_blob", "two_blobs"])
def test_emd_split_kernels_match_oracle(oracle, emd_split, cluster_width, split, width, kind):
    """The rounds after the last four-bidders-per-wave round run in emd_lean_kernel (split = 1,
    the default) or stay in emd_auction_kernel (split = 0): both must give the oracle's bits --
    assignment and distances -- for every cluster width.  `duplicates` forces value ties (tie
    order on original indices); `few_rounds_left` ends the auction a few rounds after the
    earliest possible hand-over (the hand-over needs 64 remaining rounds, so some clouds are
    handed over and some are not); `last_round_forced` ends with persons still unassigned (forced
    assignment, emd_cuda.cu:201) in the second kernel; the blobs make the search cube cover the
    grid (linear-scan fallback) and keep many bidders on few objects."""
    from mvp_benchmark_amd.metrics import emd
    emd_split(split)
    cluster_width(width)
    if kind == "uniform":
        x1, x2, eps, iters = rand_clouds(31, 3, 4096, 3), rand_clouds(32, 3, 4096, 3), 0.004, 3000
    elif kind == "duplicates":
        x1 = np.tile(rand_clouds(33, 2, 512, 3), (1, 4, 1))
        x2, eps, iters = np.tile(rand_clouds(34, 2, 256, 3), (1, 8, 1)), 0.005, 1500
    elif kind == "few_rounds_left":
        x1, x2, eps = rand_clouds(35, 2, 2048, 3), rand_clouds(36, 2, 2048, 3), 0.004
        handover = _handover_round(oracle, x1, x2, eps, width)
        assert 0 < handover < 2900
        iters = handover + 70
    elif kind == "last_round_forced":
        x1, x2, eps, iters = rand_clouds(37, 2, 2048, 3), rand_clouds(38, 2, 2048, 3), 0.002, 400
        assert oracle.emd_forward_ex(x1, x2, eps, iters)[3][:, -1].min() > 0   # persons left for the forced round
    elif kind == "person_blob":
        x1 = (0.5 + 0.01 * rand_clouds(69, 2, 1024, 3)).astype(np.float32)
        x2, eps, iters = rand_clouds(70, 2, 1024, 3), 0.004, 1500
    elif kind == "object_blob":
        x1 = rand_clouds(71, 2, 2048, 3)
        x2, eps, iters = (0.3 + 0.002 * rand_clouds(72, 2, 2048, 3)).astype(np.float32), 0.004, 800
    else:
        x1 = np.concatenate([0.2 + 0.02 * rand_clouds(73, 2, 1024, 3), 0.8 + 0.02 * rand_clouds(74, 2, 1024, 3)], 1).astype(np.float32)
        x2 = np.concatenate([0.25 + 0.02 * rand_clouds(75, 2, 1024, 3), 0.7 + 0.05 * rand_clouds(76, 2, 1024, 3)], 1).astype(np.float32)
        eps, iters = 0.004, 1200
    dist, ass = emd()(dev(x1), dev(x2), eps, iters)
    od, oa = oracle.emd_forward(x1, x2, eps, iters)
    np.testing.assert_array_equal(ass.cpu().numpy(), oa)
    np.testing.assert_array_equal(dist.cpu().numpy(), od)


def test_emd_second_kernel_really_runs(emd_split):
    """The statistics words say how many rounds ran in total; with the split on, a long auction
    must have been handed over (hand-over record: next round > 0), with it off not."""
    from mvp_benchmark_amd import _lib
    b, n = 2, 4096
    x1, x2 = dev(rand_clouds(81, b, n, 3)), dev(rand_clouds(82, b, n, 3))
    nbytes = _lib.emd_scratch_bytes(b, n)
    out = {}
    for split in (0, 1):
        emd_split(split)
        scratch = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
        dist = torch.zeros(b, n, device=DEV)
        ass = torch.zeros(b, n, dtype=torch.int32, device=DEV)
        _lib.call("mvp_emd_forward", DEV, b, n, x1, x2, dist, ass, 0.004, 3000, scratch, nbytes)
        torch.cuda.synchronize()
        stats = scratch[nbytes - b * 16:].view(torch.int64).view(b, 2).cpu().numpy()
        rec = _lib.emd_records(scratch, nbytes, b)                     # hand-over records + statistics
        out[split] = (dist.cpu().numpy(), ass.cpu().numpy(), stats, rec)
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
    np.testing.assert_array_equal(out[0][2], out[1][2])          # same rounds, same bids
    assert (out[0][3]["first_handover"] == 0).all()               # split off: nothing handed over
    assert (out[1][3]["first_handover"] > 0).all() and (out[1][3]["first_handover"] < 1500).all()   # split on: round of the hand-over
    assert (out[1][3]["unassigned"] <= 512).all()                 # kLeanCap (emd_common.h; 384 before the round-6 re-tuning)
    r1 = out[1][3]
    assert (r1["next_round"] == 0).all() and (r1["final_width"] == 8).all() and (r1["final_launch"] == 1).all()   # finished, by the clusters of 8 a batch of 2 gets


def test_emd_plan_on_the_call_is_stateless_and_bit_identical(emd_split):
    """mvp_emd_forward_plan (ABI 17): the launch plan travels with the call -- the process-wide knobs of
    mvp_emd_configure are neither read (a plan-less call runs the compiled-in defaults whatever was configured) nor
    written (the next mvp_emd_forward still sees the configured split), results are the same bits for every plan, and
    the hand-over record says which plan ran."""
    from mvp_benchmark_amd import _lib
    b, n = 2, 4096
    x1, x2 = dev(rand_clouds(83, b, n, 3)), dev(rand_clouds(84, b, n, 3))
    nbytes = _lib.emd_scratch_bytes(b, n)

    def run(entry, *plan):
        scratch = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
        dist = torch.zeros(b, n, device=DEV)
        ass = torch.zeros(b, n, dtype=torch.int32, device=DEV)
        _lib.call(entry, "cuda:0", b, n, x1, x2, dist, ass, 0.004, 3000, scratch, nbytes, *plan)   # (a str device: _lib normalises it)
        torch.cuda.synchronize()
        return dist.cpu().numpy(), ass.cpu().numpy(), _lib.emd_records(scratch, nbytes, b)

    emd_split(0)                                             # process-wide: the first kernel runs every round
    ref = run("mvp_emd_forward")
    assert (ref[2]["first_handover"] == 0).all()
    outs = [run("mvp_emd_forward_plan", None), run("mvp_emd_forward_plan", _lib.EmdPlan()),
            run("mvp_emd_forward_plan", _lib.EmdPlan(split=1, cluster=2)), run("mvp_emd_forward_plan", _lib.EmdPlan(split=3, resident_cap=8))]
    for o in outs:
        np.testing.assert_array_equal(o[0], ref[0])
        np.testing.assert_array_equal(o[1], ref[1])
        assert (o[2]["first_handover"] > 0).all()             # the plan's split, not the configured 0
    assert (outs[0][2]["final_launch"] == 3).all() and (outs[1][2]["final_launch"] == 3).all()   # defaults: split 5 -> resident tail
    assert (outs[2][2]["final_width"] == 2).all() and (outs[2][2]["final_launch"] == 1).all()
    assert (run("mvp_emd_forward")[2]["first_handover"] == 0).all()   # the process-wide knob is untouched
    with pytest.raises(_lib.MvpOpsError):
        run("mvp_emd_forward_plan", _lib.EmdPlan(cluster=3))


def test_emd_headline_cloud_matches_oracle(oracle, emd_split):
    """Two cloud pairs of the headline shape (16384 points, eps 0.004, 3000
    rounds) against the exhaustive oracle, bit for bit -- with the split into two
    kernels (hand-over around round 100) and with the first kernel alone.
    (~70 s of CPU per cloud, once.)"""
    from mvp_benchmark_amd.metrics import emd
    x1, x2 = rand_clouds(41, 2, 16384, 3), rand_clouds(42, 2, 16384, 3)
    od, oa = oracle.emd_forward(x1, x2, 0.004, 3000)
    for split in (2, 0):
        emd_split(split)
        dist, ass = emd()(dev(x1), dev(x2), 0.004, 3000)
        np.testing.assert_array_equal(ass.cpu().numpy(), oa)
        np.testing.assert_array_equal(dist.cpu().numpy(), od)


@pytest.mark.parametrize("b,split", [(64, 2), (64, 1), (61, 2), (40, 2), (33, 2)])
def test_emd_cfg4_full_batch_matches_oracle(oracle, emd_split, b, split):
    """BASELINE cfg 4 at its FULL batch: 64 clouds of 1024 points, eval setting (eps 0.004,
    3000 rounds), every cloud against the oracle (cfg 4 was only ever compared at
    B <= 3 below the headline size).  33..64 clouds -> four workgroups per cloud; clouds this small
    are not dealt out again at round 300 whatever `split` says (the records show it); the tiered
    launch is compared with the oracle at 4096 points below."""
    from mvp_benchmark_amd import _lib
    emd_split(split)
    x1, x2 = rand_clouds(91, 64, 1024, 3)[:b], rand_clouds(92, 64, 1024, 3)[:b]
    nbytes = _lib.emd_scratch_bytes(b, 1024)
    scratch = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    dist = torch.zeros(b, 1024, device=DEV)
    ass = torch.zeros(b, 1024, dtype=torch.int32, device=DEV)
    _lib.call("mvp_emd_forward", DEV, b, 1024, dev(x1), dev(x2), dist, ass, 0.004, 3000, scratch, nbytes)
    torch.cuda.synchronize()
    od, oa = oracle.emd_forward(x1, x2, 0.004, 3000)
    np.testing.assert_array_equal(ass.cpu().numpy(), oa)
    np.testing.assert_array_equal(dist.cpu().numpy(), od)
    rec = _lib.emd_records(scratch, nbytes, b)
    assert (rec["next_round"] == 0).all()                           # every cloud finished
    running = rec["first_handover"] > 0                             # handed over at all (else: done in the first kernel)
    tiered = running & (rec["final_launch"] == 2)                   # finished by the tiered launch (still running at round 300)
    assert set(rec["final_width"][running & ~tiered].tolist()) <= {4}
    assert tiered.sum() == 0


@pytest.mark.parametrize("split", [2, 1])
def test_emd_cfg4_n2048_full_batch_clustered_matches_oracle(oracle, emd_split, split):
    """BASELINE cfg 4, n = 2048 at its FULL batch on the clustered kernels (four workgroups per cloud; the default
    path, which finishes these clouds LDS-resident, is in test_gpu_emd_resident.py): every cloud against the oracle
    (2048 and 8192 points had only been compared at B <= 3, i.e. on eight workgroups per cloud)."""
    from mvp_benchmark_amd import _lib
    emd_split(split)
    b, n = 64, 2048
    x1, x2 = rand_clouds(191, b, n, 3), rand_clouds(192, b, n, 3)
    nbytes = _lib.emd_scratch_bytes(b, n)
    scratch = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    dist = torch.zeros(b, n, device=DEV)
    ass = torch.zeros(b, n, dtype=torch.int32, device=DEV)
    _lib.call("mvp_emd_forward", DEV, b, n, dev(x1), dev(x2), dist, ass, 0.004, 3000, scratch, nbytes)
    torch.cuda.synchronize()
    od, oa, ost = oracle.emd_forward(x1, x2, 0.004, 3000, return_stats=True)
    np.testing.assert_array_equal(ass.cpu().numpy(), oa)
    np.testing.assert_array_equal(dist.cpu().numpy(), od)
    rec = _lib.emd_records(scratch, nbytes, b)
    np.testing.assert_array_equal(rec["rounds"], ost[:, 0])
    np.testing.assert_array_equal(rec["bids"], ost[:, 1])
    assert (rec["next_round"] == 0).all() and (rec["final_launch"] <= 2).all()
    assert set(rec["final_width"][rec["first_handover"] > 0].tolist()) <= {4}      # below 4096 points nothing is dealt out again


def test_emd_cfg4_n8192_full_batch_tiered_equals_single_kernel_and_oracle(oracle, emd_split):
    """BASELINE cfg 4, n = 8192 at its FULL batch: the tiered launches (four workgroups per cloud to round 300, then
    8 .. 2 by load) against the first kernel running every round alone -- all 64 clouds, distances, assignments,
    rounds and bids -- and the heaviest and the lightest cloud against the exhaustive oracle (~10 s of CPU each)."""
    from mvp_benchmark_amd import _lib
    b, n = 64, 8192
    x1n, x2n = rand_clouds(193, b, n, 3), rand_clouds(194, b, n, 3)
    x1, x2 = dev(x1n), dev(x2n)
    nbytes = _lib.emd_scratch_bytes(b, n)
    out = {}
    for split in (0, 2):
        emd_split(split)
        scratch = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
        dist = torch.zeros(b, n, device=DEV)
        ass = torch.zeros(b, n, dtype=torch.int32, device=DEV)
        _lib.call("mvp_emd_forward", DEV, b, n, x1, x2, dist, ass, 0.004, 3000, scratch, nbytes)
        torch.cuda.synchronize()
        out[split] = (dist.cpu().numpy(), ass.cpu().numpy(), _lib.emd_records(scratch, nbytes, b))
    np.testing.assert_array_equal(out[0][0], out[2][0])
    np.testing.assert_array_equal(out[0][1], out[2][1])
    r0, r2 = out[0][2], out[2][2]
    np.testing.assert_array_equal(r0["rounds"], r2["rounds"])
    np.testing.assert_array_equal(r0["bids"], r2["bids"])
    tiered = r2["final_launch"] == 2
    assert tiered.sum() >= 56 and len(set(r2["final_width"][tiered].tolist())) >= 3, r2["final_width"]
    heavy, light = int(np.argmax(r2["bids"])), int(np.argmin(r2["bids"]))
    od, oa, ost = oracle.emd_forward(x1n[[heavy, light]], x2n[[heavy, light]], 0.004, 3000, return_stats=True)
    np.testing.assert_array_equal(out[2][1][[heavy, light]], oa)
    np.testing.assert_array_equal(out[2][0][[heavy, light]], od)
    np.testing.assert_array_equal(r2["bids"][[heavy, light]], ost[:, 1])


@pytest.mark.parametrize("b", [33, 40, 61])
def test_emd_tiered_widths_ragged_batches_match_oracle(oracle, b):
    """The tiered launch with cloud slots left empty (33, 40, 61 clouds in grids laid out for 40 / 40 /
    64): all clouds equal the first kernel alone, six of them (the heaviest, the lightest, four
    others) the exhaustive oracle.  4096 points: ~4 s of CPU per cloud."""
    from mvp_benchmark_amd import _lib
    n = 4096
    x1n, x2n = rand_clouds(97, b, n, 3), rand_clouds(98, b, n, 3)
    x1, x2 = dev(x1n), dev(x2n)
    nbytes = _lib.emd_scratch_bytes(b, n)
    out = {}
    try:
        for split in (0, 2):
            _lib.emd_configure(split=split)
            scratch = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
            dist = torch.zeros(b, n, device=DEV)
            ass = torch.zeros(b, n, dtype=torch.int32, device=DEV)
            _lib.call("mvp_emd_forward", DEV, b, n, x1, x2, dist, ass, 0.004, 3000, scratch, nbytes)
            torch.cuda.synchronize()
            out[split] = (dist.cpu().numpy(), ass.cpu().numpy(), _lib.emd_records(scratch, nbytes, b))
    finally:
        _lib.emd_configure(split=_lib.EMD_DEFAULT_SPLIT)
    np.testing.assert_array_equal(out[0][0], out[2][0])
    np.testing.assert_array_equal(out[0][1], out[2][1])
    rec = out[2][2]
    tiered = rec["final_launch"] == 2
    # empty slots (or fewer than 8 per XCD): the widths follow the loads' squares, from the instantiated set; they fill the grid
    wd = rec["final_width"][tiered]
    assert tiered.sum() >= b - 8 and set(wd.tolist()) <= {2, 3, 4, 5, 6, 8} and len(set(wd.tolist())) >= 3, rec["final_width"]
    assert 4 * ((b + 7) // 8 * 8) - 16 <= int(wd.sum()) <= 4 * ((b + 7) // 8 * 8), int(wd.sum())
    order = np.argsort(rec["unassigned"])
    pick = sorted(set([int(order[0]), int(order[-1]), 1, b // 3, b // 2, b - 2]))
    od, oa = oracle.emd_forward(x1n[pick], x2n[pick], 0.004, 3000)
    np.testing.assert_array_equal(out[2][1][pick], oa)
    np.testing.assert_array_equal(out[2][0][pick], od)


def test_emd_tiered_launch_refused_late_or_repeated_gives_the_same_bits():
    """A device the tiered kernel's grid does not fit on: the launcher must finish the auction with the
    fixed-width kernel from the round the first lean launch stopped at.  Simulated in child processes that load
    libmvpops_hooks.so (the release sources + -DMVP_TEST_HOOKS: the only build that reads MVP_EMD_TIERS_FAIL and the plan
    knobs, once per process); digests of the results against each other and against the release library's."""
    import hashlib, subprocess, sys, os
    code = (
        "import sys, hashlib, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from mvp_benchmark_amd import _lib\n"
        "_lib.LIB_PATH = _lib.LIB_PATH.replace('libmvpops.so', 'libmvpops_hooks.so')\n"     # the build with -DMVP_TEST_HOOKS: reads the MVP_EMD_* knobs
        "rng = np.random.default_rng(123)\n"
        "b, n = 40, 4096\n"
        "x1 = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda(); x2 = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda()\n"
        "nbytes = _lib.emd_scratch_bytes(b, n); scratch = torch.zeros(nbytes, dtype=torch.uint8, device='cuda')\n"
        "dist = torch.zeros(b, n, device='cuda'); ass = torch.zeros(b, n, dtype=torch.int32, device='cuda')\n"
        "_lib.call('mvp_emd_forward', dist.device, b, n, x1, x2, dist, ass, 0.004, 3000, scratch, nbytes); torch.cuda.synchronize()\n"
        "rec = _lib.emd_records(scratch, nbytes, b)\n"
        "print(hashlib.sha1(dist.cpu().numpy().tobytes() + ass.cpu().numpy().tobytes()).hexdigest(), sorted(set((rec['final_launch'] * 100 + rec['final_width']).tolist())))\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    # also: the plan made late (round 1500: clouds that collapsed to one workgroup are stopped and resumed), and made
    # three times (rounds 600, 1300, 2000: the tiered launch itself stops and hands over to the next one)
    # (MVP_EMD_SPLIT=2: the tiered launches; the default, 3, finishes clouds of this size LDS-resident -- last variant)
    for extra in ({"MVP_EMD_SPLIT": "2"}, {"MVP_EMD_SPLIT": "2", "MVP_EMD_TIERS_FAIL": "1"}, {"MVP_EMD_SPLIT": "2", "MVP_EMD_PLAN_ROUND": "1500"},
                  {"MVP_EMD_SPLIT": "2", "MVP_EMD_PLAN_ROUND": "600", "MVP_EMD_PLAN_EVERY": "700"}, {"MVP_EMD_SPLIT": "0"}, {}):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert len(set(o.split()[0] for o in outs)) == 1, outs
    assert "[204]" in outs[1] and "[204]" not in outs[0], outs      # refused: every cloud finished by a 4-wide launch on granule set 2
    assert "[301]" in outs[5], outs                                  # default: one workgroup per cloud, LDS-resident


def test_emd_headline_batch_tiered_equals_single_kernel_and_oracle(oracle, emd_split):
    """The headline shape itself (64 x 16384, eps 0.004, 3000 rounds): the default's three launches against the first
    kernel running every round alone -- distances, assignments, rounds and bids identical; the records show the tiers;
    the heaviest and one of the lightest clouds also against the exhaustive oracle."""
    from mvp_benchmark_amd import _lib
    b, n = 64, 16384
    g = torch.Generator().manual_seed(0)
    x1, x2 = torch.rand(b, n, 3, generator=g).to(DEV), torch.rand(b, n, 3, generator=g).to(DEV)
    nbytes = _lib.emd_scratch_bytes(b, n)
    out = {}
    for split in (0, 2):
        emd_split(split)
        scratch = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
        dist = torch.zeros(b, n, device=DEV)
        ass = torch.zeros(b, n, dtype=torch.int32, device=DEV)
        _lib.call("mvp_emd_forward", DEV, b, n, x1, x2, dist, ass, 0.004, 3000, scratch, nbytes)
        torch.cuda.synchronize()
        out[split] = (dist.clone(), ass.clone(), _lib.emd_records(scratch, nbytes, b))
    assert torch.equal(out[0][0], out[2][0]) and torch.equal(out[0][1], out[2][1])
    r0, r2 = out[0][2], out[2][2]
    assert (r0["rounds"] == r2["rounds"]).all() and (r0["bids"] == r2["bids"]).all() and (r2["rounds"] == 3000).all()
    assert (r0["first_handover"] == 0).all() and (r2["final_launch"] == 2).all()
    assert sorted(set(r2["final_width"].tolist())) == [2, 3, 4, 5, 8] and (r2["final_width"] == 8).sum() == 8
    # ... and the oracle itself for one cloud that finished on 8 workgroups and one that finished on 2 (~70 s of CPU)
    heavy = int(np.argmax(r2["unassigned"])); light = int(np.argmin(np.where(r2["final_width"] == 2, r2["unassigned"], 1 << 30)))
    assert r2["final_width"][heavy] == 8 and r2["final_width"][light] == 2
    pick = [heavy, light]
    od, oa = oracle.emd_forward(x1[pick].cpu().numpy(), x2[pick].cpu().numpy(), 0.004, 3000)
    np.testing.assert_array_equal(out[2][1][pick].cpu().numpy(), oa)
    np.testing.assert_array_equal(out[2][0][pick].cpu().numpy(), od)


def test_emd_tiered_widths_match_the_single_kernel(emd_split):
    """64 clouds of 4096 points: most are still running at round 300, where the default (split = 2)
    deals the 256 workgroups out again -- per XCD the heaviest cloud gets 8, the next 5, then 4, 4, 3, 3, 3
    and the lightest 2.  Same bits as the first kernel running every round alone (which the tests above
    pin to the oracle at this size), same statistics, and the records show the three widths."""
    from mvp_benchmark_amd import _lib
    b, n = 64, 4096
    x1, x2 = dev(rand_clouds(95, b, n, 3)), dev(rand_clouds(96, b, n, 3))
    nbytes = _lib.emd_scratch_bytes(b, n)
    out = {}
    for split in (0, 2):
        emd_split(split)
        scratch = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
        dist = torch.zeros(b, n, device=DEV)
        ass = torch.zeros(b, n, dtype=torch.int32, device=DEV)
        _lib.call("mvp_emd_forward", DEV, b, n, x1, x2, dist, ass, 0.004, 3000, scratch, nbytes)
        torch.cuda.synchronize()
        stats = scratch[nbytes - b * 16:].view(torch.int64).view(b, 2).cpu().numpy()
        out[split] = (dist.cpu().numpy(), ass.cpu().numpy(), stats, _lib.emd_records(scratch, nbytes, b))
    np.testing.assert_array_equal(out[0][0], out[2][0])
    np.testing.assert_array_equal(out[0][1], out[2][1])
    np.testing.assert_array_equal(out[0][2], out[2][2])          # same rounds, same bids
    rec = out[2][3]
    tiered = rec["final_launch"] == 2
    assert tiered.sum() >= 48, rec["final_launch"]
    w = rec["final_width"][tiered]
    assert (w == 8).sum() == 8 and (w == 5).sum() == 8 and (w == 4).sum() == 16 and set(w.tolist()) <= {2, 3, 4, 5, 8}, np.bincount(w)
    # the wider a cloud's cluster, the more persons it had unassigned at round 300
    un = rec["unassigned"][tiered]
    assert un[w == 8].min() >= un[w == 5].max() >= un[w == 5].min() >= un[w == 4].max()


@pytest.mark.parametrize("kind", ["random", "tie_heavy"])
def test_emd_result_is_a_member_of_the_reference_outcome_set(kind):
    """The HIP result against oracle/emulator.py -- a thread-by-thread replay of the reference's
    seven kernels per round, written independently of mvp_oracle.c -- under the ascending thread
    order (the legal schedule of the reference program that GetMax's race is pinned to): bit-equal,
    i.e. the HIP result IS one of the outcomes the reference program can produce
    (tests/test_emulator.py shows oracle == emulator and which launches are order-sensitive)."""
    from oracle import emulator
    from mvp_benchmark_amd.metrics import emd
    if kind == "random":
        x1, x2, eps, iters = rand_clouds(201, 2, 1024, 3), rand_clouds(202, 2, 1024, 3), 0.005, 50
    else:
        x1 = np.tile(rand_clouds(203, 1, 256, 3), (1, 4, 1))
        x2, eps, iters = np.tile(rand_clouds(204, 1, 128, 3), (1, 8, 1)), 0.005, 60
    dist, ass = emd()(dev(x1), dev(x2), eps, iters)
    ed, ea, info = emulator.emd_forward(x1, x2, eps, iters, "ascending", return_info=True)
    np.testing.assert_array_equal(ass.cpu().numpy(), ea)
    np.testing.assert_array_equal(dist.cpu().numpy(), ed)
    if kind == "tie_heavy":
        assert info[0]["racy_getmax_launches"] > 0


def test_emd_tiny_squared_distances_take_the_full_sqrt_path(oracle):
    """emd_value's square root is the correction step of the compiler's expansion, which is exact for 0 and for
    x >= 2^-96; a wave that holds a squared distance in (0, 2^-96) -- points closer than 3.5e-15 without being
    equal, only possible next to the origin -- must take the full expansion (scaling of denormal-range inputs).
    A quarter of both clouds is put on a 1e-17-spaced lattice at the origin (squared distances of 1e-34 and
    below, many of them denormal floats); assignment and distances still equal the CPU's sqrtf bit for bit."""
    from mvp_benchmark_amd.metrics import emd
    rng = np.random.default_rng(17)
    x1, x2 = rand_clouds(301, 2, 1024, 3), rand_clouds(302, 2, 1024, 3)
    x1[:, :256] = (rng.integers(0, 64, (2, 256, 3)) * 1e-17).astype(np.float32)
    x2[:, :256] = (rng.integers(0, 64, (2, 256, 3)) * 1e-17).astype(np.float32)
    d2 = ((x1[:, :256, None] - x2[:, None, :256]) ** 2).sum(-1)
    assert ((d2 > 0) & (d2 < 2.0 ** -96)).any()
    dist, ass = emd()(dev(x1), dev(x2), 0.005, 120)
    od, oa = oracle.emd_forward(x1, x2, 0.005, 120)
    np.testing.assert_array_equal(ass.cpu().numpy(), oa)
    np.testing.assert_array_equal(dist.cpu().numpy(), od)


def test_emd_cluster_widths_agree_at_full_size(cluster_width):
    """16384 points, eval setting: 1 and 4 workgroups per cloud give identical
    assignments (the oracle needs minutes at this size)."""
    from mvp_benchmark_amd.metrics import emd
    g = torch.Generator().manual_seed(5)
    x1 = torch.rand(8, 16384, 3, generator=g).to(DEV)
    x2 = torch.rand(8, 16384, 3, generator=g).to(DEV)
    out = {}
    for w in (1, 4):
        cluster_width(w)
        out[w] = emd()(x1, x2, 0.004, 3000)
    assert torch.equal(out[1][1], out[4][1]) and torch.equal(out[1][0], out[4][0])


def test_emd_cluster_under_competing_load(oracle, cluster_width):
    """The workgroups of a cluster exchange state inside one launch (write-through
    stores, L1-bypassing loads, polled granules).  Run it while another stream
    streams through HBM and occupies CUs, so that members start unevenly and
    every hand-off happens under load; repeated calls also re-use a dirty
    scratch buffer.  Results must not move by a bit."""
    from mvp_benchmark_amd.metrics import emd
    cluster_width(4)
    x1, x2 = rand_clouds(21, 4, 4096, 3), rand_clouds(22, 4, 4096, 3)
    od, oa = oracle.emd_forward(x1, x2, 0.004, 600)
    t1, t2 = dev(x1), dev(x2)
    side = torch.cuda.Stream()
    big = torch.rand(64 * 1024 * 1024, device=DEV)
    for rep in range(4):
        with torch.cuda.stream(side):
            acc = big
            for _ in range(40):
                acc = acc * 1.0001 + 0.5      # ~0.5 GB of traffic per op, thousands of workgroups
        dist, ass = emd()(t1, t2, 0.004, 600)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(ass.cpu().numpy(), oa)
        np.testing.assert_array_equal(dist.cpu().numpy(), od)


def test_emd_clustered_input_with_ties(oracle):
    """Duplicated points force equal values, i.e. the tie order."""
    from mvp_benchmark_amd.metrics import emd
    base = rand_clouds(0, 1, 256, 3)
    x2 = np.tile(base, (1, 4, 1))
    x1 = np.tile(rand_clouds(1, 1, 512, 3), (1, 2, 1))
    dist, ass = emd()(dev(x1), dev(x2), 0.005, 300)
    od, oa = oracle.emd_forward(x1, x2, 0.005, 300)
    np.testing.assert_array_equal(ass.cpu().numpy(), oa)
    np.testing.assert_array_equal(dist.cpu().numpy(), od)


def test_emd_backward_and_guards(oracle):
    from mvp_benchmark_amd.metrics import emd
    x1, x2 = rand_clouds(1, 2, 1024, 3), rand_clouds(2, 2, 1024, 3)
    t1 = dev(x1).requires_grad_()
    dist, ass = emd()(t1, dev(x2), 0.005, 50)
    g = rand_clouds(3, 2, 1024)
    (dist * dev(g)).sum().backward()
    gx = oracle.emd_backward(x1, x2, g, ass.cpu().numpy())
    np.testing.assert_allclose(t1.grad.cpu().numpy(), gx, rtol=1e-6, atol=1e-7)
    with pytest.raises(Exception):
        emd()(dev(x1[:, :1000]), dev(x2[:, :1000]), 0.005, 50)


def test_emd_lazy_status_follows_the_tensors_stream():
    """The status words of a call are copied, and their event recorded, on the stream the kernels ran on -- the
    current stream of the tensors' device (the event was recorded on the current device's default
    stream, so on another stream / device it completed at once and check() read the buffer before the copy
    landed).  A side stream kept busy by a long kernel: the filed event must not be complete yet."""
    from mvp_benchmark_amd.metrics import emd
    from mvp_benchmark_amd.metrics.EMD import emd_module
    emd_module.check(block=True)
    assert not emd_module._PENDING
    x1, x2 = dev(rand_clouds(601, 2, 1024, 3)), dev(rand_clouds(602, 2, 1024, 3))
    side = torch.cuda.Stream()
    big = torch.rand(8192, 8192, device=DEV)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(20):
            big = big @ big * 1e-4          # ~100 ms of work ahead of the auction on this stream
        dist, ass = emd()(x1, x2, 0.005, 50)
    assert len(emd_module._PENDING) == 1
    ev = emd_module._PENDING[0][0]
    assert not ev.query()                   # still behind the side stream's work (the default stream is idle)
    emd_module.check(block=False)
    assert len(emd_module._PENDING) == 1    # not examined early
    emd_module.check(block=True)            # waits for the side stream, examines, raises nothing
    assert not emd_module._PENDING
    side.synchronize()
    assert float(dist.min()) >= 0 and int(ass.min()) >= 0


def test_emd_lazy_status_is_silent_under_graph_capture():
    """Inside a stream capture nothing may be queried: forward must neither examine earlier calls nor file
    its own (the replayed graph re-runs the launch, not the bookkeeping)."""
    from mvp_benchmark_amd.metrics import emd
    from mvp_benchmark_amd.metrics.EMD import emd_module
    x1, x2 = dev(rand_clouds(611, 2, 1024, 3)), dev(rand_clouds(612, 2, 1024, 3))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            d0, a0 = emd()(x1, x2, 0.005, 50)            # warm-up on the capture stream: files entries
    s.synchronize()
    assert emd_module._PENDING
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        d1, a1 = emd()(x1, x2, 0.005, 50)
    pending = len(emd_module._PENDING)
    g.replay()
    torch.cuda.synchronize()
    assert len(emd_module._PENDING) == pending             # the capture filed nothing
    assert torch.equal(d1, d0) and torch.equal(a1, a0)
    emd_module.check(block=True)


def test_emd_full_size_self_consistency():
    """Headline shape (64, 16384), eval settings: the identity the reference's
    test_emd prints (emd_module.py:100-104) plus near-bijection."""
    from mvp_benchmark_amd.metrics import emd
    g = torch.Generator().manual_seed(0)
    x1 = torch.rand(64, 16384, 3, generator=g).to(DEV)
    x2 = torch.rand(64, 16384, 3, generator=g).to(DEV)
    dist, ass = emd()(x1, x2, 0.004, 3000)
    assert ass.min() >= 0 and ass.max() < 16384
    m = torch.gather(x2, 1, ass.long()[..., None].expand(-1, -1, 3))
    assert torch.allclose(dist, ((x1 - m) ** 2).sum(-1), rtol=1e-5, atol=1e-9)
    uniq = min(len(torch.unique(r)) for r in ass)
    assert uniq > 16384 * 0.995
    assert 0.01 < dist.sqrt().mean().item() < 0.05


# ---------------------------------------------------------------------- fps
@pytest.mark.parametrize("b,n,m", [(2, 5, 5), (3, 64, 16), (2, 100, 100), (2, 777, 200),
                                   (4, 1024, 256), (4, 2048, 2048), (3, 3072, 1536),
                                   (2, 5000, 300), (2, 16384, 2048), (1, 20000, 64)])
def test_fps_matches_oracle(oracle, b, n, m):
    from mvp_benchmark_amd.mm3d_pn2 import furthest_point_sample
    x = rand_clouds(n + m, b, n, 3)
    idx = furthest_point_sample(dev(x), m)
    assert idx.dtype == torch.int32 and tuple(idx.shape) == (b, m)
    np.testing.assert_array_equal(idx.cpu().numpy(), oracle.furthest_point_sample(x, m))


def test_fps_ties_on_lattice(oracle):
    """A lattice makes many distances exactly equal: the bit-reversed-slot tie
    rule of the reference's LDS tree decides every round."""
    from mvp_benchmark_amd.mm3d_pn2 import furthest_point_sample
    for side, m in [(4, 40), (8, 300), (11, 700)]:
        g = np.stack(np.meshgrid(*[np.arange(side) / 16.0] * 3, indexing="ij"), -1)
        x = g.reshape(1, -1, 3).astype(np.float32)
        x = np.concatenate([x, x[:, ::-1]], 0).copy()
        idx = furthest_point_sample(dev(x), m)
        np.testing.assert_array_equal(idx.cpu().numpy(), oracle.furthest_point_sample(x, m))


@pytest.mark.parametrize("b,n,m", [(3, 1025, 300), (2, 1536, 1536), (2, 1537, 100), (3, 2047, 512), (2, 3000, 1500),
                                   (2, 4096, 1024), (2, 4097, 333), (2, 6144, 2000), (2, 7000, 512), (2, 8192, 700)])
def test_fps_mid_sizes_match_oracle_and_leave_min_distances(oracle, b, n, m):
    """1024 < N <= 8192 (the register-resident kernel with 2..8 points per lane):
    indices equal the oracle's, the running minima left in `temp` are the
    distances to the nearest selected point."""
    from mvp_benchmark_amd import _lib
    x = rand_clouds(n * 3 + m, b, n, 3)
    tx = dev(x)
    temp = torch.full((b, n), 1e10, device=DEV)
    idx = torch.zeros(b, m, dtype=torch.int32, device=DEV)
    _lib.call("mvp_furthest_point_sampling", tx.device, b, n, m, tx, temp, idx)
    want = oracle.furthest_point_sample(x, m)
    np.testing.assert_array_equal(idx.cpu().numpy(), want)
    sel = np.take_along_axis(x, want[:, :-1, None].astype(np.int64), 1).astype(np.float64)     # the last pick updates nothing
    d = ((x[:, :, None, :].astype(np.float64) - sel[:, None, :, :]) ** 2).sum(-1).min(-1) if m > 1 else np.full((b, n), 1e10)
    np.testing.assert_allclose(temp.cpu().numpy(), d, rtol=1e-5, atol=1e-7)


def test_fps_mid_sizes_ties(oracle):
    """Lattices and duplicated points for 1024 < N <= 8192: most rounds have
    several points at the maximum, inside one lane and across lanes."""
    from mvp_benchmark_amd.mm3d_pn2 import furthest_point_sample
    cases = []
    for side, m in [(11, 700), (13, 900), (16, 1200), (18, 600), (20, 500)]:
        g = np.stack(np.meshgrid(*[np.arange(side) / 32.0] * 3, indexing="ij"), -1).reshape(1, -1, 3).astype(np.float32)
        perm = np.random.default_rng(side).permutation(g.shape[1])
        cases.append((np.ascontiguousarray(np.concatenate([g, g[:, perm]], 0)), m))
    cases.append((np.tile(rand_clouds(3, 2, 700, 3), (1, 4, 1)), 1000))       # 2800 points, every point 4 times
    cases.append((np.tile(rand_clouds(4, 2, 1000, 3), (1, 7, 1)), 1500))      # 7000 points, every point 7 times
    cases.append((np.zeros((2, 3000, 3), np.float32), 50))                    # all points identical
    for x, m in cases:
        idx = furthest_point_sample(dev(x), m)
        np.testing.assert_array_equal(idx.cpu().numpy(), oracle.furthest_point_sample(x, m))


@pytest.mark.parametrize("b,n,m", [(3, 4097, 500), (2, 6000, 1000), (2, 8192, 2048), (2, 12000, 700), (2, 16384, 2048)])
def test_fps_sorted_variant_matches_oracle(oracle, b, n, m):
    """The Morton-sorted kernel for 4096 < N <= 16384 (whole waves skip the
    distance update): indices and the final running minima equal the plain
    kernel's / the oracle's."""
    from mvp_benchmark_amd import _lib
    x = rand_clouds(n * 7 + m, b, n, 3)
    tx = dev(x)
    outs = []
    for name in ("mvp_furthest_point_sampling", "mvp_furthest_point_sampling_sorted"):
        temp = torch.full((b, n), 1e10, device=DEV)
        idx = torch.zeros(b, m, dtype=torch.int32, device=DEV)
        if name.endswith("sorted"):
            nbytes = _lib.fps_scratch_bytes(b, n)
            ws = torch.full((nbytes,), 0xCD, dtype=torch.uint8, device=DEV)
            _lib.call(name, tx.device, b, n, m, tx, temp, idx, ws, nbytes)
        else:
            _lib.call(name, tx.device, b, n, m, tx, temp, idx)
        outs.append((idx, temp))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    np.testing.assert_array_equal(outs[1][0].cpu().numpy(), oracle.furthest_point_sample(x, m))


def test_fps_sorted_variant_ties(oracle):
    """Lattice and duplicated points in the sorted kernel's size range: most
    rounds have several points at the maximum, so the original-index tie rule
    (bit-reversed slot, then lowest index) decides."""
    from mvp_benchmark_amd import _lib
    g = np.stack(np.meshgrid(*[np.arange(18) / 32.0] * 3, indexing="ij"), -1).reshape(1, -1, 3).astype(np.float32)
    lattice = np.ascontiguousarray(np.concatenate([g, g[:, np.random.default_rng(0).permutation(g.shape[1])]], 0))   # 5832 points
    dup = np.tile(rand_clouds(3, 2, 1500, 3), (1, 4, 1))                                                              # 6000 points
    for x, m in [(lattice, 900), (dup, 1600)]:
        b, n = x.shape[:2]
        tx = dev(x)
        temp = torch.full((b, n), 1e10, device=DEV)
        idx = torch.zeros(b, m, dtype=torch.int32, device=DEV)
        nbytes = _lib.fps_scratch_bytes(b, n)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        _lib.call("mvp_furthest_point_sampling_sorted", tx.device, b, n, m, tx, temp, idx, ws, nbytes)
        np.testing.assert_array_equal(idx.cpu().numpy(), oracle.furthest_point_sample(x, m))


@pytest.mark.parametrize("b,n,m,w,kind", [(3, 8192, 400, 2, "random"), (3, 8192, 400, 4, "random"), (2, 16384, 300, 4, "random"),
                                           (2, 4096, 512, 4, "lattice"), (2, 2048, 2048, 2, "lattice"), (5, 6144, 64, 2, "random")])
def test_fps_cluster_variant_matches_oracle(oracle, b, n, m, w, kind):
    """mvp_furthest_point_sampling_cluster: w workgroups per cloud, local winners exchanged through
    memory every round -- same indices and the same final min-distance array as the oracle, ties
    included (lattice: the maximum is attained many times in every round, inside one thread's
    points and across members; m = n: every point is sampled)."""
    from mvp_benchmark_amd import _lib
    if kind == "random":
        x = rand_clouds(n + w, b, n, 3)
    else:
        side = round(n ** (1 / 3)) if round(n ** (1 / 3)) ** 3 == n else 16
        g = np.stack(np.meshgrid(*[np.arange(side, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3) / side
        x = np.tile(g[None], (b, (n + len(g) - 1) // len(g), 1))[:, :n].astype(np.float32)
    nbytes = _lib.fps_cluster_scratch_bytes(b)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    temp = torch.full((b, n), 1e10, device=DEV)
    idx = torch.zeros(b, m, dtype=torch.int32, device=DEV)
    _lib.call("mvp_furthest_point_sampling_cluster", DEV, b, n, m, w, dev(x), temp, idx, scratch, nbytes)
    np.testing.assert_array_equal(idx.cpu().numpy(), oracle.furthest_point_sample(x, m))
    # the plain kernel's temp is the reference's: same array
    temp2 = torch.full((b, n), 1e10, device=DEV)
    idx2 = torch.zeros(b, m, dtype=torch.int32, device=DEV)
    _lib.call("mvp_furthest_point_sampling", DEV, b, n, m, dev(x), temp2, idx2)
    assert torch.equal(temp, temp2)


def test_fps_equals_the_grid_emulator():
    """The HIP FPS against the thread-by-thread replay of furthest_point_sampling_kernel
    (oracle/emulator.py: strided per-thread scan, shared-memory tree level by level) on a lattice
    with ties in almost every round and on a random cloud whose size is not a power of two."""
    from oracle import emulator
    from mvp_benchmark_amd.mm3d_pn2 import furthest_point_sample
    g = np.stack(np.meshgrid(*[np.arange(8, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3) / 8
    lattice = np.tile(g[None], (2, 3, 1)).astype(np.float32)            # (2, 1536, 3), every point three times
    for x, m in ((lattice, 96), (rand_clouds(77, 2, 1500, 3), 128)):
        got = furthest_point_sample(dev(x), m).cpu().numpy()
        np.testing.assert_array_equal(got, emulator.furthest_point_sample(x, m, "ascending"))


def test_fps_with_dist_matches_oracle(oracle):
    from mvp_benchmark_amd.mm3d_pn2 import furthest_point_sample_with_dist
    for n, m in [(50, 20), (300, 100), (1500, 64)]:
        x = rand_clouds(n, 2, n, 8)
        d = ((x[:, :, None] - x[:, None]) ** 2).sum(-1).astype(np.float32)
        idx = furthest_point_sample_with_dist(dev(d), m)
        np.testing.assert_array_equal(idx.cpu().numpy(), oracle.furthest_point_sample_with_dist(d, m))
        # the matrix the reference's F-FPS really feeds it (|a|^2 + |b|^2 - 2ab in float32) has slightly
        # negative entries on and near the diagonal: they must order below every positive distance
        sq = (x * x).sum(-1)
        d2 = (sq[:, :, None] + sq[:, None] - 2 * np.einsum("bnc,bmc->bnm", x, x)).astype(np.float32)
        d2[:, np.arange(n), np.arange(n)] = -np.abs(d2[:, np.arange(n), np.arange(n)]) - 1e-7
        idx = furthest_point_sample_with_dist(dev(d2), m)
        np.testing.assert_array_equal(idx.cpu().numpy(), oracle.furthest_point_sample_with_dist(d2, m))


# ------------------------------------------------------- ball_query/knn/3nn
def test_ball_query_matches_oracle(oracle):
    from mvp_benchmark_amd.mm3d_pn2 import ball_query
    # ECG get_uniform_loss shapes (model_utils.py:205-211) and generic ones
    for n, m, r0, r1, s in [(1024, 51, 0.0, 0.0632, 4), (2048, 102, 0.0, 0.1095, 24),
                            (500, 40, 0.05, 0.2, 8), (3000, 257, 0.0, 0.3, 64), (70, 3, 0.0, 5.0, 100)]:
        xyz = rand_clouds(n, 3, n, 3)
        ctr = np.ascontiguousarray(xyz[:, :m]).copy()
        ctr[:, 0] += 50.0  # a centre with no hit
        idx = ball_query(r0, r1, s, dev(xyz), dev(ctr))
        assert idx.dtype == torch.int32
        np.testing.assert_array_equal(idx.cpu().numpy(), oracle.ball_query(r0, r1, s, xyz, ctr))


@pytest.mark.parametrize("k,n,m", [(1, 100, 100), (10, 2048, 2048), (16, 3000, 500),
                                   (20, 1024, 1500), (33, 700, 130), (100, 1000, 70)])
def test_knn_matches_oracle(oracle, k, n, m):
    from mvp_benchmark_amd.mm3d_pn2 import knn
    xyz, ctr = rand_clouds(k, 2, n, 3), rand_clouds(k + 1, 2, m, 3)
    idx = knn(k, dev(xyz), dev(ctr), False)
    assert tuple(idx.shape) == (2, k, m) and idx.dtype == torch.int32
    np.testing.assert_array_equal(idx.cpu().numpy(), oracle.knn(k, xyz, ctr))
    # duplicated points: the heap's handling of equal distances
    xyz2 = np.concatenate([xyz[:, : n // 2], xyz[:, : n // 2]], 1)
    idx = knn(k, dev(xyz2), dev(ctr), False)
    np.testing.assert_array_equal(idx.cpu().numpy(), oracle.knn(k, xyz2, ctr))


@pytest.mark.parametrize("k,n,m,kind", [(16, 4096, 4096, "random"), (1, 5000, 1024, "random"), (32, 4100, 1500, "random"),
                                        (9, 8192, 3000, "duplicates"), (16, 4096, 4096, "lattice"), (16, 6000, 2048, "clustered"),
                                        (16, 16384, 16384, "self"), (20, 3072, 3072, "self"), (10, 2048, 2048, "duplicates"),
                                        (16, 2048, 2048, "random"), (20, 2048, 2048, "lattice")])
def test_knn_sorted_variant_is_bit_identical(oracle, k, n, m, kind):
    """mvp_knn_sorted (clouds of >= 4096 candidates, square searches from 2048 points: Morton-sorted, pruned) against the oracle's replay of the
    reference's heap (knn_cuda.cu:58-95), indices AND distances, through the C ABI: random clouds (the k + 1 nearest
    pairwise different: the pruned search alone decides), duplicated points and a lattice (equal distances: those
    queries are recomputed wave by wave with the reference's heap sequence -- the counters say how many), ragged sizes (padding of the sorted
    sets), clustered queries far from most candidates, and self-kNN at the headline cloud size."""
    from mvp_benchmark_amd import _lib
    b = 2 if n <= 8192 else 1
    xyz = rand_clouds(700 + k, b, n, 3)
    ctr = rand_clouds(701 + k, b, m, 3)
    if kind == "duplicates":
        xyz = np.concatenate([xyz[:, : n // 2], xyz[:, : n // 2]], 1)
    elif kind == "lattice":
        side = (8, 16, 16) if n == 2048 else (16, 16, 16)              # (round 4: square searches from 2048 points take this route)
        g = np.stack(np.meshgrid(*[np.arange(d) for d in side], indexing="ij"), -1).reshape(-1, 3).astype(np.float32) / 16
        xyz = np.stack([g[np.random.default_rng(s).permutation(n)] for s in range(b)])
        ctr = xyz.copy()
    elif kind == "clustered":
        ctr = (0.9 + 0.05 * ctr).astype(np.float32)
    elif kind == "self":
        ctr = xyz.copy()
    want_i, want_d = oracle.knn(k, xyz, ctr, return_dist=True)        # (b, k, m) indices, (b, m, k) squared distances
    nbytes = _lib.knn_scratch_bytes(b, n, m)
    scratch = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    idx = torch.zeros(b, m, k, dtype=torch.int32, device=DEV)
    d2 = torch.zeros(b, m, k, device=DEV)
    _lib.call("mvp_knn_sorted", DEV, b, n, m, k, dev(xyz), dev(ctr), idx, d2, scratch, nbytes)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(idx.cpu().numpy().transpose(0, 2, 1), want_i)
    np.testing.assert_array_equal(d2.cpu().numpy(), want_d)
    # the counters of the fix-up pass (queries recomputed by the reference's own heap sequence) sit in front of the id lists
    cnt_at = nbytes - ((b * m * 4 + 15) // 16 * 16) - ((b * 4 + 15) // 16 * 16)
    recomputed = scratch[cnt_at: cnt_at + b * 4].view(torch.int32).cpu().numpy()
    if kind in ("random", "clustered", "self"):
        assert recomputed.sum() <= 2               # (two equal distances among 17 random ones: ~1e-5 per query)
    elif kind == "lattice":
        assert (recomputed == m).all()             # every query of a lattice has equidistant neighbours
    else:
        assert recomputed.sum() > 0
    # the operator takes the same route for these sizes
    from mvp_benchmark_amd.mm3d_pn2 import knn
    np.testing.assert_array_equal(knn(k, dev(xyz), dev(ctr), False).cpu().numpy(), want_i)


def test_knn_default_centre_and_transposed(oracle):
    from mvp_benchmark_amd.mm3d_pn2 import knn
    xyz = rand_clouds(5, 2, 600, 3)
    a = knn(4, dev(xyz))
    b = knn(4, dev(xyz.transpose(0, 2, 1)), None, True)
    assert torch.equal(a, b)
    np.testing.assert_array_equal(a.cpu().numpy(), oracle.knn(4, xyz))


@pytest.mark.parametrize("n,m", [(768, 384), (1536, 768), (3072, 1536), (256, 64), (100, 2), (33, 1)])
def test_three_nn_matches_oracle(oracle, n, m):
    from mvp_benchmark_amd.mm3d_pn2 import three_nn
    tgt, src = rand_clouds(n, 3, n, 3), rand_clouds(m, 3, m, 3)
    dist, idx = three_nn(dev(tgt), dev(src))
    od, oi = oracle.three_nn(tgt, src)
    np.testing.assert_array_equal(idx.cpu().numpy(), oi)
    np.testing.assert_array_equal(dist.cpu().numpy(), od)


# ------------------------------------------------- gather/group/interpolate
@pytest.mark.parametrize("b,c,n,m", [(2, 3, 3072, 1536), (3, 64, 1536, 15360), (1, 13, 100, 7), (2, 1, 5, 300),
                                     (2, 20, 5000, 9000), (2, 9, 8192, 4000), (1, 4, 9000, 3000), (2, 17, 2048, 3073)])
def test_gather_points_fwd_bwd(oracle, b, c, n, m):
    from mvp_benchmark_amd.mm3d_pn2 import gather_points
    f = rand_clouds(0, b, c, n)
    idx = np.random.default_rng(1).integers(0, n, (b, m)).astype(np.int32)
    tf = dev(f).requires_grad_()
    out = gather_points(tf, dev(idx))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), oracle.gather_points(f, idx))
    g = rand_clouds(2, b, c, m)
    out.backward(dev(g))
    np.testing.assert_allclose(tf.grad.cpu().numpy(), oracle.gather_points_grad(g, idx, n), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("b,c,n,p,s", [(2, 3, 2048, 102, 24), (2, 64, 1536, 768, 1), (1, 10, 50, 7, 5),
                                       (2, 80, 768, 768, 16), (2, 12, 3072, 3072, 16)])
def test_grouping_operation_fwd_bwd(oracle, b, c, n, p, s):
    from mvp_benchmark_amd.mm3d_pn2 import grouping_operation
    f = rand_clouds(0, b, c, n)
    idx = np.random.default_rng(1).integers(0, n, (b, p, s)).astype(np.int32)
    tf = dev(f).requires_grad_()
    out = grouping_operation(tf, dev(idx))
    assert tuple(out.shape) == (b, c, p, s)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), oracle.grouping_operation(f, idx))
    g = rand_clouds(2, b, c, p, s)
    out.backward(dev(g))
    np.testing.assert_allclose(tf.grad.cpu().numpy(), oracle.grouping_operation_grad(g, idx, n), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("b,c,m,n", [(2, 512, 384, 768), (2, 128, 1536, 3072), (1, 5, 10, 33), (2, 9, 1024, 3100),
                                     (1, 3, 9000, 500)])
def test_three_interpolate_fwd_bwd(oracle, b, c, m, n):
    from mvp_benchmark_amd.mm3d_pn2 import three_interpolate
    f = rand_clouds(0, b, c, m)
    idx = np.random.default_rng(1).integers(0, m, (b, n, 3)).astype(np.int32)
    w = rand_clouds(3, b, n, 3)
    tf = dev(f).requires_grad_()
    out = three_interpolate(tf, dev(idx), dev(w))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), oracle.three_interpolate(f, idx, w))
    g = rand_clouds(2, b, c, n)
    out.backward(dev(g))
    np.testing.assert_allclose(tf.grad.cpu().numpy(), oracle.three_interpolate_grad(g, idx, w, m), rtol=1e-5, atol=1e-5)


def test_scatter_gradients_hub_graph_and_plain_entry_points(oracle):
    """Every entry scatters into one of three destinations (the longest possible
    lists of the transposed index), and the *_ws entry points agree with the
    plain ones they fall back to."""
    from mvp_benchmark_amd import _lib
    b, c, n, p, s = 2, 11, 1500, 700, 9
    rng = np.random.default_rng(4)
    idx = rng.integers(0, 3, (b, p, s)).astype(np.int32) * 700
    g = rand_clouds(5, b, c, p, s)
    want = oracle.grouping_operation_grad(g, idx, n)
    tg, ti = dev(g), dev(idx)
    plain = torch.zeros(b, c, n, device=DEV)
    _lib.call("mvp_group_points_grad", DEV, b, c, n, p, s, tg, ti, plain)
    nbytes = _lib.scatter_scratch_bytes(b, n, p * s, 1)
    assert nbytes > 0
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    ws = torch.zeros(b, c, n, device=DEV)
    _lib.call("mvp_group_points_grad_ws", DEV, b, c, n, p, s, tg, ti, ws, scratch, nbytes, 0)
    nows = torch.zeros(b, c, n, device=DEV)
    _lib.call("mvp_group_points_grad_ws", DEV, b, c, n, p, s, tg, ti, nows, None, 0, 0)
    for got in (plain, ws, nows):
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-5, atol=2e-4)
    # accumulate-into contract (mode 0): a second call doubles the result -- here on the index the
    # first call left in the scratch (MVP_SCATTER_INDEX_READY)
    _lib.call("mvp_group_points_grad_ws", DEV, b, c, n, p, s, tg, ti, ws, scratch, nbytes, 2)
    np.testing.assert_allclose(ws.cpu().numpy(), 2 * want, rtol=2e-5, atol=4e-4)
    # MVP_SCATTER_OVERWRITE: the destination's contents do not matter, with and without scratch
    for sc, nb, mode in ((scratch, nbytes, 1 | 2), (scratch, nbytes, 1), (None, 0, 1)):
        dirty = torch.full((b, c, n), 7.5, device=DEV)
        _lib.call("mvp_group_points_grad_ws", DEV, b, c, n, p, s, tg, ti, dirty, sc, nb, mode)
        np.testing.assert_allclose(dirty.cpu().numpy(), want, rtol=2e-5, atol=2e-4)
    assert _lib.scatter_scratch_bytes(b, 8193, 100, 1) == 0 and _lib.scatter_scratch_bytes(b, 100, 100, 2) == 0


@pytest.mark.parametrize("b,c,n,p,k,kind", [(2, 37, 700, 300, 16, "random"), (3, 128, 3072, 1536, 16, "random"),
                                            (2, 24, 1024, 256, 10, "ties"), (1, 5, 30000, 100, 4, "long_rows"),
                                            (2, 64, 64, 64, 1, "k1")])
def test_gather_max_matches_oracle_composition(oracle, b, c, n, p, k, kind):
    """mvp_gather_max (neighbour max-pool of edge_preserve_sampling, fused) against the oracle's
    gather followed by NumPy's max / first-argmax: values exact, the recorded winner = the FIRST
    maximal neighbour (ties: quantised / ReLU-like features), gradient = scatter of grad_out to
    the winners (float atomics in LDS: 1e-6).  `long_rows`: a row does not fit the LDS staging
    buffer, the Python wrapper takes the gather + reduce route."""
    from mvp_benchmark_amd import _lib
    from mvp_benchmark_amd.mm3d_pn2.functional import gather_max
    rng = np.random.default_rng(b * 1000 + c)
    f = rand_clouds(c, b, c, n)
    if kind == "ties":
        f = np.maximum(np.round(f * 4) / 4 - 0.5, 0).astype(np.float32)      # many exact zeros and repeats
    idx = rng.integers(0, n, (b, p, k)).astype(np.int32)
    gathered = oracle.gather_points(f, idx.reshape(b, p * k)).reshape(b, c, p, k)
    want = gathered.max(-1)
    jstar = gathered.argmax(-1)                                             # first maximum
    want_arg = np.take_along_axis(np.broadcast_to(idx[:, None], (b, c, p, k)), jstar[..., None], -1)[..., 0]
    tf = dev(f).requires_grad_()
    out = gather_max(tf, dev(idx))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), want)
    go = rand_clouds(7, b, c, p)
    out.backward(dev(go))
    wg = np.zeros((b, c, n), np.float32)
    bi, ci, _ = np.meshgrid(np.arange(b), np.arange(c), np.arange(p), indexing="ij")
    np.add.at(wg, (bi, ci, want_arg), go)
    np.testing.assert_allclose(tf.grad.cpu().numpy(), wg, rtol=1e-6, atol=1e-6)
    if kind != "long_rows":
        # the C ABI directly: the winner array, and accumulate vs overwrite
        o2 = torch.empty(b, c, p, device=DEV)
        a2 = torch.empty(b, c, p, dtype=torch.int32, device=DEV)
        _lib.call("mvp_gather_max", DEV, b, c, n, p, k, dev(f), dev(idx), o2, a2)
        np.testing.assert_array_equal(a2.cpu().numpy(), want_arg)
        acc = torch.full((b, c, n), 2.0, device=DEV)
        _lib.call("mvp_gather_max_grad", DEV, b, c, n, p, dev(go), a2, acc, 0)
        np.testing.assert_allclose(acc.cpu().numpy(), wg + 2.0, rtol=1e-6, atol=1e-6)
        _lib.call("mvp_gather_max_grad", DEV, b, c, n, p, dev(go), a2, acc, 1)
        np.testing.assert_allclose(acc.cpu().numpy(), wg, rtol=1e-6, atol=1e-6)
    else:
        with pytest.raises(_lib.MvpOpsError):
            _lib.call("mvp_gather_max", DEV, b, c, n, p, k, dev(f), dev(idx), torch.empty(b, c, p, device=DEV),
                      torch.empty(b, c, p, dtype=torch.int32, device=DEV))


def test_gather_max_propagates_nan_like_torch_max():
    """A NaN among the neighbours gives NaN (and the first NaN as the winner), a row of -inf keeps -inf and the
    first neighbour -- torch.max's rules, i.e. what the gather_points + torch.max route of the same wrapper
    (rows longer than the LDS strip, op_config gather_max = False) and the reference (model_utils.py:101-104) give."""
    from mvp_benchmark_amd import _lib
    b, c, n, p, k = 2, 5, 300, 64, 8
    rng = np.random.default_rng(3)
    f = rand_clouds(21, b, c, n)
    f[0, 1, ::7] = np.nan
    f[1, 2, :] = -np.inf
    f[1, 3, 5] = np.nan
    idx = rng.integers(0, n, (b, p, k)).astype(np.int32)
    idx[1, 0, 3] = 5
    out = torch.empty(b, c, p, device=DEV)
    arg = torch.empty(b, c, p, dtype=torch.int32, device=DEV)
    _lib.call("mvp_gather_max", DEV, b, c, n, p, k, dev(f), dev(idx), out, arg)
    tf = torch.tensor(f)
    g = torch.gather(tf.unsqueeze(2).expand(b, c, p, n), 3, torch.tensor(idx).long().unsqueeze(1).expand(b, c, p, k))
    want, j = g.max(-1)
    np.testing.assert_array_equal(out.cpu().numpy(), want.numpy())            # (NaN == NaN under assert_array_equal)
    want_arg = torch.gather(torch.tensor(idx).long().unsqueeze(1).expand(b, c, p, k), 3, j.unsqueeze(-1))[..., 0]
    np.testing.assert_array_equal(arg.cpu().numpy(), want_arg.numpy())
    assert torch.isnan(out[1, 3, 0]) and int(arg[1, 3, 0]) == 5


def test_scatter_gradient_index_cache(oracle):
    """The autograd Functions keep the inverted index of an index tensor and reuse
    it (several gathers through one neighbour graph; a retained graph
    differentiated twice); an in-place edit of the index invalidates it."""
    from mvp_benchmark_amd.mm3d_pn2 import functional as F, gather_points, three_interpolate
    F.clear_scatter_cache()
    b, c, n, m = 2, 9, 700, 2100
    rng = np.random.default_rng(1)
    idx = dev(rng.integers(0, n, (b, m)).astype(np.int32))
    f1 = dev(rand_clouds(1, b, c, n)).requires_grad_()
    f2 = dev(rand_clouds(2, b, 3, n)).requires_grad_()
    g1, g2 = rand_clouds(3, b, c, m), rand_clouds(4, b, 3, m)
    o1, o2 = gather_points(f1, idx), gather_points(f2, idx)
    (o1 * dev(g1)).sum().add((o2 * dev(g2)).sum()).backward()
    assert len(F._TRANSPOSED) == 1                                   # one index, sorted once, used twice
    np.testing.assert_allclose(f1.grad.cpu().numpy(), oracle.gather_points_grad(g1, idx.cpu().numpy(), n), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(f2.grad.cpu().numpy(), oracle.gather_points_grad(g2, idx.cpu().numpy(), n), rtol=1e-5, atol=1e-5)
    idx[:, :100] = 0                                                 # in-place edit: the cached lists are stale
    f1.grad = None
    (gather_points(f1, idx) * dev(g1)).sum().backward()
    assert len(F._TRANSPOSED) == 2
    np.testing.assert_allclose(f1.grad.cpu().numpy(), oracle.gather_points_grad(g1, idx.cpu().numpy(), n), rtol=1e-5, atol=1e-5)
    # three_interpolate: the weights are part of the key
    i3 = dev(rng.integers(0, n, (b, 900, 3)).astype(np.int32))
    w = dev(rand_clouds(5, b, 900, 3))
    go = rand_clouds(6, b, c, 900)
    f1.grad = None
    (three_interpolate(f1, i3, w) * dev(go)).sum().backward()
    want = oracle.three_interpolate_grad(go, i3.cpu().numpy(), w.cpu().numpy(), n)
    np.testing.assert_allclose(f1.grad.cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    w2 = w * 2
    f1.grad = None
    (three_interpolate(f1, i3, w2) * dev(go)).sum().backward()
    np.testing.assert_allclose(f1.grad.cpu().numpy(), 2 * want, rtol=1e-5, atol=2e-5)
    F.clear_scatter_cache()


def test_scatter_gradient_index_cache_hazards(oracle, monkeypatch):
    """(1) a backward whose kernel call raises must not leave a cache entry
    behind (the next backward would reduce over an unbuilt list); (2) an index buffer
    rewritten through a raw pointer is invisible to `_version`: `clear_scatter_cache()`
    (or SCATTER_CACHE = False) is the documented way out."""
    from mvp_benchmark_amd import _lib
    from mvp_benchmark_amd.mm3d_pn2 import functional as F, gather_points
    F.clear_scatter_cache()
    b, c, n, m = 2, 5, 600, 1800
    rng = np.random.default_rng(7)
    idx = dev(rng.integers(0, n, (b, m)).astype(np.int32))
    f = dev(rand_clouds(1, b, c, n)).requires_grad_()
    g = rand_clouds(2, b, c, m)
    # (1) first backward fails inside the library call
    real_call = F.call
    def failing(name, *a, **k):
        if name.endswith("_grad_ws"):
            raise _lib.MvpOpsError("injected")
        return real_call(name, *a, **k)
    monkeypatch.setattr(F, "call", failing)
    with pytest.raises(_lib.MvpOpsError):
        (gather_points(f, idx) * dev(g)).sum().backward()
    assert len(F._TRANSPOSED) == 0
    monkeypatch.setattr(F, "call", real_call)
    f.grad = None
    (gather_points(f, idx) * dev(g)).sum().backward()
    assert len(F._TRANSPOSED) == 1
    np.testing.assert_allclose(f.grad.cpu().numpy(), oracle.gather_points_grad(g, idx.cpu().numpy(), n), rtol=1e-5, atol=1e-5)
    # (2) the index buffer refilled behind PyTorch's back (same pointer, same _version)
    new_idx = rng.integers(0, n, (b, m)).astype(np.int32)
    idx.data.copy_(torch.from_numpy(new_idx))          # .data: no version bump
    F.clear_scatter_cache()
    f.grad = None
    (gather_points(f, idx) * dev(g)).sum().backward()
    np.testing.assert_allclose(f.grad.cpu().numpy(), oracle.gather_points_grad(g, new_idx, n), rtol=1e-5, atol=1e-5)
    # opt-out: nothing is cached at all
    F.clear_scatter_cache()
    monkeypatch.setattr(F, "SCATTER_CACHE", False)
    f.grad = None
    (gather_points(f, idx) * dev(g)).sum().backward()
    assert len(F._TRANSPOSED) == 0
    np.testing.assert_allclose(f.grad.cpu().numpy(), oracle.gather_points_grad(g, new_idx, n), rtol=1e-5, atol=1e-5)


def test_query_and_group_composition(oracle):
    """QueryAndGroup / GroupAll (group_points.py:11-163) = ball_query or knn
    -> group xyz - centre (/ radius) (+) group features, composed in NumPy
    from the oracle ops."""
    from mvp_benchmark_amd.mm3d_pn2 import QueryAndGroup, GroupAll
    xyz, feats = rand_clouds(0, 2, 400, 3), rand_clouds(1, 2, 6, 400)
    ctr = np.ascontiguousarray(xyz[:, :50])
    out = QueryAndGroup(0.3, 8, normalize_xyz=True)(dev(xyz), dev(ctr), dev(feats))
    idx = oracle.ball_query(0, 0.3, 8, xyz, ctr)
    gx = oracle.grouping_operation(np.ascontiguousarray(xyz.transpose(0, 2, 1)), idx)
    gx = (gx - ctr.transpose(0, 2, 1)[..., None]) / np.float32(0.3)
    want = np.concatenate([gx, oracle.grouping_operation(feats, idx)], 1)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-6, atol=1e-7)
    out = QueryAndGroup(None, 5)(dev(xyz), dev(ctr), dev(feats))
    idx = np.ascontiguousarray(oracle.knn(5, xyz, ctr).transpose(0, 2, 1))
    gx = oracle.grouping_operation(np.ascontiguousarray(xyz.transpose(0, 2, 1)), idx) - ctr.transpose(0, 2, 1)[..., None]
    want = np.concatenate([gx, oracle.grouping_operation(feats, idx)], 1)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-6, atol=1e-7)
    ga = GroupAll()(dev(xyz), None, dev(feats))
    assert tuple(ga.shape) == (2, 9, 1, 400)


def test_points_sampler_feature_fps_and_fs_values(oracle):
    """F-FPS / FS (points_sampler.py:119-158): FPS on the squared distance of the
    joint [xyz, feature] vectors (calc_square_dist, utils.py:4-31, norm=False);
    FS = [F-FPS, D-FPS].  The distance matrix against float64 NumPy, the indices
    against the oracle's furthest_point_sample_with_dist on that matrix."""
    from mvp_benchmark_amd.mm3d_pn2 import Points_Sampler
    from mvp_benchmark_amd.mm3d_pn2.modules import calc_square_dist
    xyz, feats = rand_clouds(7, 2, 300, 3), rand_clouds(8, 2, 5, 300)
    joint = np.concatenate([xyz, feats.transpose(0, 2, 1)], 2)
    dmat = calc_square_dist(dev(joint), dev(joint), norm=False)
    j64 = joint.astype(np.float64)
    want = ((j64[:, :, None] - j64[:, None]) ** 2).sum(-1)
    np.testing.assert_allclose(dmat.cpu().numpy(), want, rtol=1e-4, atol=2e-5)
    ffps = oracle.furthest_point_sample_with_dist(dmat.cpu().numpy(), 40)
    dfps = oracle.furthest_point_sample(xyz, 40)
    got = Points_Sampler([40], ['F-FPS'], [-1])(dev(xyz), dev(feats))
    np.testing.assert_array_equal(got.cpu().numpy(), ffps)
    got = Points_Sampler([40], ['FS'], [-1])(dev(xyz), dev(feats))
    np.testing.assert_array_equal(got.cpu().numpy(), np.concatenate([ffps, dfps], 1))
    # two ranges, F-FPS on the first 128 points, FS on the rest (indices offset by the range start)
    got = Points_Sampler([16, 8], ['F-FPS', 'FS'], [128, -1])(dev(xyz), dev(feats)).cpu().numpy()
    d0 = calc_square_dist(dev(joint[:, :128]), dev(joint[:, :128]), norm=False).cpu().numpy()
    d1 = calc_square_dist(dev(joint[:, 128:]), dev(joint[:, 128:]), norm=False).cpu().numpy()
    want = np.concatenate([oracle.furthest_point_sample_with_dist(d0, 16),
                           oracle.furthest_point_sample_with_dist(d1, 8) + 128,
                           oracle.furthest_point_sample(np.ascontiguousarray(xyz[:, 128:]), 8) + 128], 1)
    np.testing.assert_array_equal(got, want)


def test_query_and_group_uniform_sample(oracle):
    """uniform_sample / return_unique_cnt (group_points.py:80-92): a row keeps its
    unique neighbours (sorted, as torch.unique returns them) and fills the other
    slots with draws from them; the count is returned.  GroupAll values."""
    from mvp_benchmark_amd.mm3d_pn2 import GroupAll, QueryAndGroup
    xyz, feats = rand_clouds(2, 2, 300, 3), rand_clouds(3, 2, 4, 300)
    ctr = np.ascontiguousarray(xyz[:, :40])
    S = 12
    torch.manual_seed(0)
    out, gxyz, cnt = QueryAndGroup(0.12, S, uniform_sample=True, return_unique_cnt=True,
                                   return_grouped_xyz=True)(dev(xyz), dev(ctr), dev(feats))
    out, gxyz, cnt = out.cpu().numpy(), gxyz.cpu().numpy(), cnt.cpu().numpy()
    idx = oracle.ball_query(0, 0.12, S, xyz, ctr)                     # padded with the first hit
    assert out.shape == (2, 7, 40, S) and cnt.shape == (2, 40)
    partial_rows = 0
    for b in range(2):
        for p in range(40):
            uniq = np.unique(idx[b, p])
            assert cnt[b, p] == len(uniq)
            partial_rows += len(uniq) < S
            np.testing.assert_array_equal(out[b, 3:, p, :len(uniq)], feats[b][:, uniq])
            np.testing.assert_allclose(gxyz[b, :, p, :len(uniq)], xyz[b][uniq].T - ctr[b, p][:, None], rtol=0, atol=1e-7)
            for sidx in range(len(uniq), S):                            # the draws: each one of the unique neighbours
                assert (np.abs(feats[b][:, uniq] - out[b, 3:, p, sidx][:, None]).max(0) == 0).any()
            np.testing.assert_array_equal(out[b, :3, p], gxyz[b, :, p])
    assert partial_rows > 20                                            # the resampling branch really ran
    ga = GroupAll()(dev(xyz), None, dev(feats)).cpu().numpy()
    np.testing.assert_array_equal(ga[:, :3, 0], xyz.transpose(0, 2, 1))
    np.testing.assert_array_equal(ga[:, 3:, 0], feats)
    np.testing.assert_array_equal(GroupAll(use_xyz=False)(dev(xyz), None, dev(feats)).cpu().numpy()[:, :, 0], feats)


def test_points_sampler(oracle):
    from mvp_benchmark_amd.mm3d_pn2 import Points_Sampler
    xyz, feats = rand_clouds(0, 2, 512, 3), rand_clouds(1, 2, 4, 512)
    idx = Points_Sampler([32, 16], ['D-FPS', 'D-FPS'], [256, -1])(dev(xyz), dev(feats))
    a = oracle.furthest_point_sample(np.ascontiguousarray(xyz[:, :256]), 32)
    b = oracle.furthest_point_sample(np.ascontiguousarray(xyz[:, 256:]), 16) + 256
    np.testing.assert_array_equal(idx.cpu().numpy(), np.concatenate([a, b], 1))
    idx = Points_Sampler([24], ['FS'], [-1])(dev(xyz), dev(feats))
    assert tuple(idx.shape) == (2, 48)


def _surface_clouds(n, seed):
    """Three surface-shaped cloud pairs of n points (MVP's clouds are samples of 2-manifolds,
    completion/dataset.py:21-34): a sphere against independent samples of it, a torus against itself + noise
    0.03, the faces of a box against itself + noise 0.01; in [0, 1]^3."""
    g = torch.Generator().manual_seed(seed)

    def sphere():
        v = torch.randn(n, 3, generator=g)
        return 0.5 + 0.4 * v / v.norm(dim=1, keepdim=True)

    def torus():
        u, v = 2 * math.pi * torch.rand(n, generator=g), 2 * math.pi * torch.rand(n, generator=g)
        return torch.stack([0.5 + (0.3 + 0.12 * torch.cos(v)) * torch.cos(u), 0.5 + (0.3 + 0.12 * torch.cos(v)) * torch.sin(u),
                            0.5 + 0.12 * torch.sin(v)], 1)

    def box():
        p = torch.rand(n, 3, generator=g)
        face = torch.randint(0, 6, (n,), generator=g)
        p.scatter_(1, (face % 3).unsqueeze(1), (face // 3).float().unsqueeze(1))
        return 0.15 + 0.7 * p

    gt = torch.stack([sphere(), torus(), box()])
    t, b = gt[1], gt[2]
    pred = torch.stack([sphere(), (t + 0.03 * torch.randn(n, 3, generator=g)).clamp(0, 1),
                        (b + 0.01 * torch.randn(n, 3, generator=g)).clamp(0, 1)])
    return pred.numpy().astype(np.float32), gt.numpy().astype(np.float32)


@pytest.mark.parametrize("n,split", [(8192, 5), (16384, 5), (2048, 5), (2048, 2)])
def test_emd_surface_shaped_clouds_match_oracle(oracle, emd_split, n, split):
    """every EMD-vs-oracle case at the headline size was a pair of uniform volumes.  Surface-shaped
    clouds put 30-200 objects into an occupied cell of the search grid (the chunked cell lists of round 4) and run
    auctions of another shape: independent samples of a surface need ~1.4x the bids of two volumes, ground truth +
    noise a quarter of them in as many rounds (profiles/r4_emd_surfaces.txt).  Bit for bit against the exhaustive
    oracle, default launch sequence (gathered-bid rounds; LDS-resident tail at 2048 points) and the tiers alone."""
    from mvp_benchmark_amd import _lib
    x1, x2 = _surface_clouds(n, 7 + n)
    emd_split(split)
    b = x1.shape[0]
    nbytes = _lib.emd_scratch_bytes(b, n)
    scratch = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    dist = torch.zeros(b, n, device=DEV)
    ass = torch.zeros(b, n, dtype=torch.int32, device=DEV)
    _lib.call("mvp_emd_forward", DEV, b, n, dev(x1), dev(x2), dist, ass, 0.004, 3000, scratch, nbytes)
    torch.cuda.synchronize()
    od, oa = oracle.emd_forward(x1, x2, 0.004, 3000)
    np.testing.assert_array_equal(ass.cpu().numpy(), oa)
    np.testing.assert_array_equal(dist.cpu().numpy(), od)
    rec = _lib.emd_records(scratch, nbytes, b)
    assert (rec["next_round"] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("b,n,m", [(3, 77, 130), (2, 2048, 4100), (1, 1, 5)])
def test_chamfer_forward_writes_every_output_element(oracle, b, n, m):
    """chamfer_3DFunction allocates its outputs with torch.empty: the kernels must overwrite all of them (both launch
    paths), whatever the buffers held."""
    import torch
    from mvp_benchmark_amd._lib import call, chamfer_scratch_bytes
    a, c = rand_clouds(n, b, n, 3), rand_clouds(m + 1, b, m, 3)
    xa, xc = torch.from_numpy(a).cuda(), torch.from_numpy(c).cuda()
    want = oracle.chamfer_forward(a, c)
    for sorted_path in (False, True):
        d1 = torch.full((b, n), float("nan"), device="cuda")
        d2 = torch.full((b, m), float("nan"), device="cuda")
        i1 = torch.full((b, n), -7, dtype=torch.int32, device="cuda")
        i2 = torch.full((b, m), -7, dtype=torch.int32, device="cuda")
        if sorted_path:
            nb = chamfer_scratch_bytes(b, n, m)
            scratch = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
            call("mvp_chamfer_forward_sorted", xa.device, b, n, m, xa, xc, d1, d2, i1, i2, scratch, nb)
        else:
            call("mvp_chamfer_forward", xa.device, b, n, m, xa, xc, d1, d2, i1, i2)
        for got, w in zip((d1, d2, i1, i2), want):
            np.testing.assert_array_equal(got.cpu().numpy(), w)
