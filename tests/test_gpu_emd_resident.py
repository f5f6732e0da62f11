"""GPU parity tests of the LDS-resident tail of the EMD auction (csrc/emd_resident.hip): clouds of at most
4096 points finish on one workgroup with the whole auction state in LDS.  Everything is compared with the
exhaustive CPU oracle bit for bit (assignment, distances, rounds, bids) through the C ABI
(utils/metrics/EMD/emd_cuda.cu:95-226)."""
import numpy as np
import pytest
import torch

from conftest import rand_clouds

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a):
    return torch.tensor(a, device=DEV)


@pytest.fixture
def knobs():
    """mvp_emd_configure for one test; the defaults come back afterwards."""
    from mvp_benchmark_amd import _lib
    yield _lib.emd_configure
    _lib.emd_configure(cluster=0, split=_lib.EMD_DEFAULT_SPLIT, resident_cap=16)


def _run(x1, x2, eps, iters):
    """mvp_emd_forward through the C ABI -> dist, assignment, records."""
    from mvp_benchmark_amd import _lib
    b, n = x1.shape[:2]
    nbytes = _lib.emd_scratch_bytes(b, n)
    scratch = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    dist = torch.zeros(b, n, device=DEV)
    ass = torch.zeros(b, n, dtype=torch.int32, device=DEV)
    _lib.call("mvp_emd_forward", DEV, b, n, dev(x1), dev(x2), dist, ass, eps, iters, scratch, nbytes)
    torch.cuda.synchronize()
    return dist.cpu().numpy(), ass.cpu().numpy(), _lib.emd_records(scratch, nbytes, b)


def _check(oracle, x1, x2, eps, iters, expect_resident=None):
    d, a, rec = _run(x1, x2, eps, iters)
    od, oa, ost = oracle.emd_forward(x1, x2, eps, iters, return_stats=True)
    np.testing.assert_array_equal(a, oa)
    np.testing.assert_array_equal(d, od)
    np.testing.assert_array_equal(rec["rounds"], ost[:, 0])     # same rounds ...
    np.testing.assert_array_equal(rec["bids"], ost[:, 1])       # ... and the same number of bids
    assert (rec["next_round"] == 0).all()
    if expect_resident is not None:
        assert ((rec["final_launch"] == 3) == expect_resident).all(), rec["final_launch"]
        assert (rec["final_width"][rec["final_launch"] == 3] == 1).all()
    return rec


@pytest.mark.parametrize("split", [5, 4, 3])
@pytest.mark.parametrize("b,n", [(3, 1024), (3, 2048), (2, 3072), (3, 4096)])
def test_resident_tail_matches_oracle(oracle, knobs, b, n, split):
    """Eval setting (eps 0.004, 3000 rounds) at every cloud size the resident rounds cover (two
    instantiations: <= 2048 and <= 4096 points), fused into the lean launch (split 4; 5, the default: with gathered-bid rounds before; member 0 of
    the cloud's cluster goes on at once) and in a launch of their own (split 3)."""
    knobs(split=split)
    x1, x2 = rand_clouds(1000 + n, b, n, 3), rand_clouds(2000 + n, b, n, 3)
    rec = _check(oracle, x1, x2, 0.004, 3000, expect_resident=True)
    assert (rec["unassigned"] <= 16).all() and (rec["unassigned"] > 0).all()


@pytest.mark.parametrize("cap", [1, 7, 16, 33, 64])
def test_resident_cap_changes_no_bit(oracle, knobs, cap):
    """The hand-over point is a tuning knob: 1 person left (the latest possible), fewer than waves, one per wave,
    several list positions per wave (33: some waves three, 64: four each -- the capacity)."""
    knobs(resident_cap=cap)
    x1, x2 = rand_clouds(301, 3, 2048, 3), rand_clouds(302, 3, 2048, 3)
    rec = _check(oracle, x1, x2, 0.004, 3000)
    assert (rec["unassigned"][rec["final_launch"] == 3] <= cap).all()
    if cap >= 7:
        assert (rec["final_launch"] == 3).all()


@pytest.mark.parametrize("split", [5, 4, 3])
@pytest.mark.parametrize("width", [1, 2, 4, 8])
def test_resident_after_every_cluster_width(oracle, knobs, width, split):
    """The resident rounds pick up whatever lists the clustered rounds left: 1, 2, 4 or 8 of them."""
    knobs(cluster=width, split=split)
    x1, x2 = rand_clouds(311, 2, 2048, 3), rand_clouds(312, 2, 2048, 3)
    _check(oracle, x1, x2, 0.004, 3000, expect_resident=True)


@pytest.mark.parametrize("kind", ["duplicates", "lattice", "few_rounds_left", "last_round_forced", "person_blob",
                                  "object_blob", "two_blobs", "noisy_copy", "short"])
def test_resident_edge_cases(oracle, kind):
    """`duplicates` / `lattice`: equal values (tie order on original indices, emd_cuda.cu:108-118,139-154,163-171)
    and equal increments inside the 1e-6 GetMax band (:188); `few_rounds_left`: the auction ends a few rounds after
    the hand-over needs its 32 rounds; `last_round_forced`: persons still unassigned in the forced last round
    (:201) inside the resident kernel; blobs: many bidders for all rounds (some clouds never reach the cap and stay
    in the clustered kernels), the seeds' filter covers every block; `noisy_copy`: prediction = ground truth + noise,
    the auction ends early; `short`: fewer rounds than any hand-over needs."""
    n, b, eps, iters = 2048, 2, 0.004, 1500
    if kind == "duplicates":
        x1 = np.tile(rand_clouds(33, 2, 512, 3), (1, 4, 1))
        x2, eps = np.tile(rand_clouds(34, 2, 256, 3), (1, 8, 1)), 0.005
    elif kind == "lattice":
        g = np.stack(np.meshgrid(*[np.arange(16)] * 3, indexing="ij"), -1).reshape(-1, 3)[:2048].astype(np.float32) / 16
        x1 = np.stack([g, g[::-1]]).astype(np.float32)
        x2 = np.stack([g[np.random.default_rng(5).permutation(2048)] + np.float32(1 / 32), g]).astype(np.float32)
    elif kind == "few_rounds_left":
        x1, x2 = rand_clouds(35, 2, 2048, 3), rand_clouds(36, 2, 2048, 3)
        trace = oracle.emd_forward_ex(x1, x2, eps, 3000)[3]
        iters = max(int(np.argmax(row <= 16)) for row in trace) + 40
    elif kind == "last_round_forced":
        x1, x2, eps, iters = rand_clouds(37, 2, 2048, 3), rand_clouds(38, 2, 2048, 3), 0.002, 400
        assert oracle.emd_forward_ex(x1, x2, eps, iters)[3][:, -1].min() > 0
    elif kind == "person_blob":
        x1 = (0.5 + 0.01 * rand_clouds(69, 2, 1024, 3)).astype(np.float32)
        x2 = rand_clouds(70, 2, 1024, 3)
    elif kind == "object_blob":
        x1 = rand_clouds(71, 2, 2048, 3)
        x2, iters = (0.3 + 0.002 * rand_clouds(72, 2, 2048, 3)).astype(np.float32), 800
    elif kind == "two_blobs":
        x1 = np.concatenate([0.2 + 0.02 * rand_clouds(73, 2, 1024, 3), 0.8 + 0.02 * rand_clouds(74, 2, 1024, 3)], 1).astype(np.float32)
        x2 = np.concatenate([0.25 + 0.02 * rand_clouds(75, 2, 1024, 3), 0.7 + 0.05 * rand_clouds(76, 2, 1024, 3)], 1).astype(np.float32)
        iters = 1200
    elif kind == "noisy_copy":
        x2 = rand_clouds(77, 2, 4096, 3)
        x1 = (x2 + 0.01 * (rand_clouds(78, 2, 4096, 3) - 0.5)).astype(np.float32)
        iters = 3000
    else:
        x1, x2, iters = rand_clouds(79, 2, 1024, 3), rand_clouds(80, 2, 1024, 3), 90
    _check(oracle, x1, x2, eps, iters)


@pytest.mark.parametrize("when", ["solo_starts_in_the_forced_last_round", "forced_last_round_inside_the_chain", "chain_ends_on_a_free_object",
                                  "full"])
def test_resident_single_bidder_chain(oracle, when):
    """One bidder left (emd_resident.h, MVP_RES_SOLO): the wave that holds it runs the remaining rounds as one chain of
    evictions without barriers.  The round counts come from the oracle's trace of unassigned persons per round: the
    auction is cut off in the very round the chain starts (it runs the forced last round, emd_cuda.cu:201-212, at once),
    a few rounds into it, is left to end on a free object, and runs the full 3000 rounds; rounds and bids are compared too."""
    x1, x2 = rand_clouds(611, 6, 1024, 3), rand_clouds(612, 6, 1024, 3)
    trace = oracle.emd_forward_ex(x1, x2, 0.004, 3000)[3]
    ones = [int(np.argmax(row == 1)) for row in trace if (row == 1).any()]
    assert len(ones) >= 2, "the seeds no longer give a single-bidder tail: pick others"
    ended = [c for c in range(len(trace)) if (trace[c] == 1).any() and trace[c][-1] == 0]
    if when == "solo_starts_in_the_forced_last_round":
        for r in sorted(set(ones))[:2]:
            for d in (0, 1, 2):   # (whichever way the trace counts: the chain's first round is one of these)
                _check(oracle, x1, x2, 0.004, r + d)
        return
    if when == "forced_last_round_inside_the_chain":
        for r in sorted(set(ones))[:3]:
            _check(oracle, x1, x2, 0.004, r + 7)
        return
    if when == "chain_ends_on_a_free_object":
        assert ended, "no cloud of these seeds converges from a single bidder: pick others"
    rec = _check(oracle, x1, x2, 0.004, 3000, expect_resident=True)
    assert (rec["rounds"] <= 3000).all()


@pytest.mark.parametrize("n", [1024, 2048])
def test_resident_cfg4_full_batch_matches_oracle(oracle, n):
    """BASELINE cfg 4 at its FULL batch through the default path: 64 clouds of 1024 / 2048 points, eval
    setting, every cloud against the oracle; every cloud ends in the resident launch."""
    x1, x2 = rand_clouds(91 + n, 64, n, 3), rand_clouds(92 + n, 64, n, 3)
    _check(oracle, x1, x2, 0.004, 3000, expect_resident=True)


def test_resident_equals_clustered_at_4096_full_batch(oracle, knobs):
    """64 clouds of 4096 points: the resident path (default: fused) against the tiered clustered launches
    (split = 2) and against the resident launch of its own (split = 3), bit for bit; the heaviest and the lightest
    cloud also against the oracle."""
    x1, x2 = rand_clouds(401, 64, 4096, 3), rand_clouds(402, 64, 4096, 3)
    d3, a3, r3 = _run(x1, x2, 0.004, 3000)
    knobs(split=3)
    d3b, a3b, r3b = _run(x1, x2, 0.004, 3000)
    np.testing.assert_array_equal(a3, a3b)
    np.testing.assert_array_equal(d3, d3b)
    np.testing.assert_array_equal(r3["bids"], r3b["bids"])
    knobs(split=2)
    d2, a2, r2 = _run(x1, x2, 0.004, 3000)
    np.testing.assert_array_equal(a3, a2)
    np.testing.assert_array_equal(d3, d2)
    np.testing.assert_array_equal(r3["rounds"], r2["rounds"])
    np.testing.assert_array_equal(r3["bids"], r2["bids"])
    assert (r3["final_launch"] == 3).all() and (r2["final_launch"] != 3).all()
    order = np.argsort(r3["bids"])
    pick = [int(order[0]), int(order[-1])]
    od, oa = oracle.emd_forward(x1[pick], x2[pick], 0.004, 3000)
    np.testing.assert_array_equal(a3[pick], oa)
    np.testing.assert_array_equal(d3[pick], od)


def test_resident_through_the_operator_with_gradient(oracle):
    """The Python operator (metrics.emd) on the default path: forward bits and the backward pass."""
    from mvp_benchmark_amd.metrics import emd
    x1, x2 = rand_clouds(501, 2, 2048, 3), rand_clouds(502, 2, 2048, 3)
    t1 = dev(x1).requires_grad_(True)
    dist, ass = emd()(t1, dev(x2), 0.004, 3000)
    od, oa = oracle.emd_forward(x1, x2, 0.004, 3000)
    np.testing.assert_array_equal(ass.cpu().numpy(), oa)
    np.testing.assert_array_equal(dist.detach().cpu().numpy(), od)
    dist.sum().backward()
    g = oracle.emd_backward(x1, x2, np.ones_like(od), oa)
    np.testing.assert_allclose(t1.grad.cpu().numpy(), g, rtol=1e-6, atol=1e-7)
