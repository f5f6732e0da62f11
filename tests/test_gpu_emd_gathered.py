"""GPU parity tests of the gathered-bid rounds of the EMD auction (csrc/emd_lean.hip, mvp_emd_configure(split = 5),
the default): once at most 256 persons of a cloud are unassigned the bids of a round travel as tagged records, every
workgroup of the cloud's cluster settles every bid itself and the round has one cluster-wide wait.  Everything is
compared with the exhaustive CPU oracle bit for bit (assignment, distances, rounds, bids) through the C ABI, and with
the plain rounds (split = 2) on full batches (utils/metrics/EMD/emd_cuda.cu:95-226)."""
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
    _lib.emd_configure(cluster=0, same_xcd=1, split=_lib.EMD_DEFAULT_SPLIT, resident_cap=16)


def _run(x1, x2, eps, iters):
    from mvp_benchmark_amd import _lib
    b, n = x1.shape[:2]
    nbytes = _lib.emd_scratch_bytes(b, n)
    scratch = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    dist = torch.zeros(b, n, device=DEV)
    ass = torch.zeros(b, n, dtype=torch.int32, device=DEV)
    _lib.call("mvp_emd_forward", DEV, b, n, x1 if torch.is_tensor(x1) else dev(x1), x2 if torch.is_tensor(x2) else dev(x2),
              dist, ass, eps, iters, scratch, nbytes)
    torch.cuda.synchronize()
    return dist.cpu().numpy(), ass.cpu().numpy(), _lib.emd_records(scratch, nbytes, b)


def _check(oracle, x1, x2, eps, iters):
    d, a, rec = _run(x1, x2, eps, iters)
    od, oa, ost = oracle.emd_forward(x1, x2, eps, iters, return_stats=True)
    np.testing.assert_array_equal(a, oa)
    np.testing.assert_array_equal(d, od)
    np.testing.assert_array_equal(rec["rounds"], ost[:, 0])
    np.testing.assert_array_equal(rec["bids"], ost[:, 1])
    assert (rec["next_round"] == 0).all()
    return rec


@pytest.mark.parametrize("b,n,width", [(2, 8192, 8), (3, 4096, 4), (2, 2048, 2), (1, 16384, 8), (5, 3072, 0)])
def test_gathered_rounds_match_oracle(oracle, knobs, b, n, width):
    """Eval setting (eps 0.004, 3000 rounds) on clusters of 8 / 4 / 2 workgroups: the rounds between the hand-over
    from the first kernel and the end (or the LDS-resident tail of the small clouds) run with gathered bids."""
    knobs(cluster=width, split=5)
    x1, x2 = rand_clouds(100 + n // 1024 + width, b, n, 3), rand_clouds(200 + n // 1024 + width, b, n, 3)
    rec = _check(oracle, x1, x2, 0.004, 3000)
    assert (rec["gathered_rounds"] > 0).all(), rec["gathered_rounds"]


def test_gathered_rounds_agent_scope_stores(oracle, knobs):
    """same_xcd = 0: the bid records, member words and state stores of the gathered-bid rounds are written through
    (the path clusters that span XCDs take)."""
    knobs(cluster=8, same_xcd=0, split=5)
    x1, x2 = rand_clouds(77, 2, 4096, 3), rand_clouds(78, 2, 4096, 3)
    rec = _check(oracle, x1, x2, 0.004, 3000)
    assert (rec["gathered_rounds"] > 0).all()


@pytest.mark.parametrize("iters", [130, 260, 401])
def test_gathered_forced_last_round(oracle, knobs, iters):
    """The auction's last round (every bidder takes its object, emd_cuda.cu:196-215) falls into the gathered-bid
    rounds: short auctions on 8192 points."""
    knobs(split=5)
    x1, x2 = rand_clouds(5 + iters, 2, 8192, 3), rand_clouds(6 + iters, 2, 8192, 3)
    rec = _check(oracle, x1, x2, 0.004, iters)
    if iters > 200:
        assert (rec["gathered_rounds"] > 0).all(), rec["gathered_rounds"]


def test_gathered_rounds_contested_objects(oracle, knobs):
    """Duplicated points: many bidders hold equal values, bid for the same object with equal increments (the 1e-6
    band of GetMax, emd_cuda.cu:181-194) -- the contests the gathered-bid rounds settle from the round's records."""
    knobs(split=5)
    rng = np.random.default_rng(11)
    b, n = 3, 2048
    x1 = np.repeat(rng.random((b, n // 4, 3), dtype=np.float32), 4, axis=1)
    x2 = np.repeat(rng.random((b, n // 2, 3), dtype=np.float32), 2, axis=1)
    for eps, iters in ((0.002, 3000), (0.008, 700)):
        rec = _check(oracle, x1, x2, eps, iters)
        assert (rec["gathered_rounds"] > 0).all()


@pytest.mark.parametrize("n", [16384, 4096])
def test_gathered_full_batch_equals_plain_rounds(knobs, n):
    """64 clouds (the headline batch at 16384 points; 4096 points: lean launch, tiered launch and LDS-resident tail
    all take part): the default equals split = 2 -- bid atomics and two all-gathers per round -- in every bit,
    round and bid, and every cloud ran gathered-bid rounds."""
    g = torch.Generator().manual_seed(3)
    x1 = torch.rand(64, n, 3, generator=g).to(DEV)
    x2 = torch.rand(64, n, 3, generator=g).to(DEV)
    out = {}
    for split in (2, 5):
        knobs(split=split)
        out[split] = _run(x1, x2, 0.004, 3000)
    np.testing.assert_array_equal(out[2][0], out[5][0])
    np.testing.assert_array_equal(out[2][1], out[5][1])
    np.testing.assert_array_equal(out[2][2]["rounds"], out[5][2]["rounds"])
    np.testing.assert_array_equal(out[2][2]["bids"], out[5][2]["bids"])
    assert (out[5][2]["gathered_rounds"] > 0).all() and (out[2][2]["gathered_rounds"] == 0).all()


def test_cfg4_n8192_full_batch_default_path_matches_oracle(oracle, knobs):
    """BASELINE cfg 4, n = 8192 at its FULL batch on the DEFAULT launch sequence (split 5: tiered widths + gathered-bid
    rounds) -- until round 5 covered only transitively (split 2 == split 0 == oracle at 8192; split 5 == split 2 at
    16384 / 4096).  The four heaviest and the four lightest clouds against the exhaustive oracle directly (~10 s of CPU
    each, OpenMP over the clouds), all 64 against the first kernel running every round alone."""
    from mvp_benchmark_amd import _lib
    b, n = 64, 8192
    x1n, x2n = rand_clouds(195, b, n, 3), rand_clouds(196, b, n, 3)
    x1, x2 = dev(x1n), dev(x2n)
    knobs(split=_lib.EMD_DEFAULT_SPLIT)
    assert _lib.EMD_DEFAULT_SPLIT == 5
    d5, a5, r5 = _run(x1, x2, 0.004, 3000)
    assert (r5["gathered_rounds"] > 0).all() and (r5["next_round"] == 0).all()
    assert (r5["final_launch"] == 2).sum() >= 56 and len(set(r5["final_width"][r5["final_launch"] == 2].tolist())) >= 3
    order = np.argsort(r5["bids"])
    pick = np.concatenate([order[:4], order[-4:]])
    od, oa, ost = oracle.emd_forward(x1n[pick], x2n[pick], 0.004, 3000, return_stats=True)
    np.testing.assert_array_equal(a5[pick], oa)
    np.testing.assert_array_equal(d5[pick], od)
    np.testing.assert_array_equal(r5["rounds"][pick], ost[:, 0])
    np.testing.assert_array_equal(r5["bids"][pick], ost[:, 1])
    knobs(split=0)
    d0, a0, r0 = _run(x1, x2, 0.004, 3000)
    np.testing.assert_array_equal(a5, a0)
    np.testing.assert_array_equal(d5, d0)
    np.testing.assert_array_equal(r5["bids"], r0["bids"])


# ---------------------------------------------------------------- the rounds of at most 16 bidders (emd_lean_round_few.inc)
def _near_pair(seed, b, n, noise):
    """A prediction near its ground truth (what a trained completion network hands to the metric): most persons find
    their object in the first rounds, the auction is down to a few bidders for most of its 3000 rounds."""
    gt = rand_clouds(seed, b, n, 3)
    pred = (gt + np.float32(noise) * (rand_clouds(seed + 1, b, n, 3) - np.float32(0.5))).astype(np.float32)
    return pred, gt


_ORACLE_RUNS = {}


def _oracle_run(oracle, key, x1, x2, eps, iters):
    """The oracle's full result (dist, assignment, stats, bidders per round) of one input, computed once per module
    (emd_forward_ex with the pinned GetMax policy IS emd_forward plus the trace: ~9 s for two clouds of 16384 points)."""
    if key not in _ORACLE_RUNS:
        _ORACLE_RUNS[key] = oracle.emd_forward_ex(x1, x2, eps, iters)
    return _ORACLE_RUNS[key]


def _check_against(run, x1, x2, eps, iters):
    d, a, rec = _run(x1, x2, eps, iters)
    od, oa, ost, _ = run
    np.testing.assert_array_equal(a, oa)
    np.testing.assert_array_equal(d, od)
    np.testing.assert_array_equal(rec["rounds"], ost[:, 0])
    np.testing.assert_array_equal(rec["bids"], ost[:, 1])
    assert (rec["next_round"] == 0).all()


@pytest.mark.parametrize("split", [5, 2])
@pytest.mark.parametrize("b,n,noise", [(2, 8192, 0.08), (2, 16384, 0.05), (3, 6144, 0.1)])
def test_few_bidder_rounds_match_oracle(oracle, knobs, b, n, noise, split):
    """Clouds above the LDS-resident tail's 4096 points that get down to <= 16 unassigned persons: member 0 of the
    cluster finishes the auction with the few-bidders rounds, entered from the gathered-bid rounds (split 5: the owner map
    is already in LDS) and from the plain rounds (split 2: it is read from the objects' records).  Bits, rounds, bids."""
    knobs(split=split)
    x1, x2 = _near_pair(700 + n // 1024, b, n, noise)
    run = _oracle_run(oracle, ("few", b, n, noise), x1, x2, 0.004, 3000)
    trace = run[3]
    assert ((trace <= 16) & (trace > 0)).sum(1).min() > 200, "these seeds no longer give a long few-bidders tail"
    _check_against(run, x1, x2, 0.004, 3000)


@pytest.mark.parametrize("extra", [1, 2, 3, 9, 40])
def test_few_bidder_rounds_cut_off(oracle, extra):
    """The auction is cut off 1 / 2 / 3 / 9 / 40 rounds after the first cloud is down to 16 persons (round ~270: before the
    launch boundary at round 300, the last one behind it): the few-bidders rounds are not entered at all, run the forced
    last round (emd_cuda.cu:201-212) as their first, or a few rounds before it."""
    x1, x2 = _near_pair(720, 2, 8192, 0.08)
    trace = _oracle_run(oracle, ("cut", 720), x1, x2, 0.004, 3000)[3]
    r16 = min(int(np.argmax(row <= 16)) for row in trace)
    assert 0 < r16 < 2900
    _check(oracle, x1, x2, 0.004, r16 + extra)


def test_few_bidder_rounds_with_equal_values_and_contests(oracle):
    """Duplicated points: equal values (the reference's tie order on original indices), several bidders for one object
    with increments inside GetMax's 1e-6 band (emd_cuda.cu:188), bucket collisions of the bid counters -- all inside the
    few-bidders rounds (8192 points: no resident tail)."""
    for seed, m, k in ((733, 2048, 4), (735, 4096, 2)):
        pred, gt = _near_pair(seed, 2, m, 0.1)
        x1, x2 = np.tile(pred, (1, k, 1)), np.tile(gt, (1, k, 1))
        run = _oracle_run(oracle, ("dups", seed), x1, x2, 0.004, 2000)
        assert ((run[3] <= 16) & (run[3] > 0)).sum(1).min() > 500
        _check_against(run, x1, x2, 0.004, 2000)
