"""The oracle against a second, independently written emulator of the reference's CUDA grids
(oracle/emulator.py: explicit blockIdx / threadIdx loops, shared-memory phases, scan trees, the
order of the threads inside a phase chosen by the test).  VERDICT r2 item 8: nothing the reference
holds pins EMD / FPS, so the single-author risk of mvp_oracle.c is reduced by agreement with a
thread-level replay of the `.cu` text -- and the emulator shows WHICH launches of the reference are
order-sensitive (GetMax with two bidders inside its 1e-6 band; nothing else)."""
import numpy as np
import pytest

import oracle
from oracle import emulator


def _rand(seed, *shape):
    return np.random.default_rng(seed).random(shape, dtype=np.float32)


def _tie_heavy():
    """Duplicated points on both sides: equal values, equal increments, several bidders per object
    inside GetMax's band in the same round."""
    x1 = np.tile(_rand(11, 1, 256, 3), (1, 4, 1))
    x2 = np.tile(_rand(12, 1, 128, 3), (1, 8, 1))
    return x1, x2, 0.005, 60


CASES = {
    "random_1024": lambda: (_rand(1, 2, 1024, 3), _rand(2, 2, 1024, 3), 0.005, 50),
    "random_2048_two_blocks": lambda: (_rand(3, 1, 2048, 3), _rand(4, 1, 2048, 3), 0.004, 40),
    "coarse_eps_forced_last_round": lambda: (_rand(5, 1, 1024, 3), _rand(6, 1, 1024, 3), 0.0005, 25),
    "tie_heavy": _tie_heavy,
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_emd_oracle_equals_grid_emulator_under_the_pinned_schedule(case):
    """Ascending thread order = what the oracle (and the HIP kernels) pin GetMax's race to: the
    last writer among the in-band bidders is the highest person index.  Bit-equal assignment and
    distances; the descending order is the oracle's other extreme (getmax_lowest)."""
    x1, x2, eps, iters = CASES[case]()
    d, a, info = emulator.emd_forward(x1, x2, eps, iters, "ascending", return_info=True)
    od, oa = oracle.emd_forward(x1, x2, eps, iters)
    np.testing.assert_array_equal(a, oa)
    np.testing.assert_array_equal(d, od)
    d2, a2 = emulator.emd_forward(x1, x2, eps, iters, "descending")
    ld, la = oracle.emd_forward_ex(x1, x2, eps, iters, getmax_lowest=True)[:2]
    np.testing.assert_array_equal(a2, la)
    np.testing.assert_array_equal(d2, ld)
    if case == "coarse_eps_forced_last_round":
        assert info[0]["unassigned"][-1] > 0          # the forced last round really assigns someone
    if case == "tie_heavy":
        assert info[0]["racy_getmax_launches"] > 0    # the race is exercised


def test_emd_only_getmax_is_order_sensitive():
    """Random thread orders in EVERY phase of every kernel (scan trees, the atomicAdd slots of
    calc_unass_idx, Bid's chunk leaders, Assign): on inputs where no two bidders meet inside
    GetMax's band the result does not move by a bit -- so nothing but GetMax depends on the
    schedule -- and on the tie-heavy input different orders give different (valid) outcomes, of
    which the pinned ascending one is the oracle's."""
    x1, x2, eps, iters = CASES["random_1024"]()
    ref = emulator.emd_forward(x1[:1], x2[:1], eps, iters, "ascending")
    for seed in range(3):
        d, a, info = emulator.emd_forward(x1[:1], x2[:1], eps, iters, np.random.default_rng(seed), return_info=True)
        assert info[0]["racy_getmax_launches"] == 0
        np.testing.assert_array_equal(a, ref[1])
        np.testing.assert_array_equal(d, ref[0])
    x1, x2, eps, iters = _tie_heavy()
    pinned = emulator.emd_forward(x1, x2, eps, iters, "ascending")
    outcomes = set()
    for seed in range(4):
        d, a = emulator.emd_forward(x1, x2, eps, iters, np.random.default_rng(100 + seed))
        outcomes.add(a.tobytes())
        # every member of the outcome set is a proper result: indices in range, distances consistent
        assert a.min() >= 0 and a.max() < x1.shape[1]
        m = np.take_along_axis(x2, a[..., None].astype(np.int64), axis=1)
        np.testing.assert_allclose(d, ((x1 - m) ** 2).sum(-1), rtol=1e-5, atol=1e-9)
    assert len(outcomes | {pinned[1].tobytes()}) > 1


def _lattice(n_side, reps):
    g = np.stack(np.meshgrid(*[np.arange(n_side, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3) / n_side
    return np.tile(g[None], (1, reps, 1)).astype(np.float32)


@pytest.mark.parametrize("n,m,kind", [(1500, 100, "random"), (513, 60, "random"), (1024, 128, "random"),
                                      (1000, 80, "lattice"), (2048, 64, "lattice_dup"), (7, 7, "random")])
def test_fps_oracle_equals_grid_emulator(n, m, kind):
    """furthest_point_sampling_kernel replayed thread by thread (strided per-thread scan with strict
    `>`, then the shared-memory tree level by level) == the oracle's closed form of the same tie rule
    (first maximum in k inside a thread, smallest bit-reversed slot across threads); block sizes 4,
    512 and 1024, lattices where almost every round has ties.  No phase is racy: any thread order
    gives the same indices."""
    if kind == "random":
        x = _rand(n, 2, n, 3)
    elif kind == "lattice":
        x = _lattice(10, 1)[:, :n]
    else:
        x = _lattice(8, 4)[:, :n]
    want = oracle.furthest_point_sample(x, m)
    np.testing.assert_array_equal(emulator.furthest_point_sample(x, m, "ascending"), want)
    np.testing.assert_array_equal(emulator.furthest_point_sample(x[:1], m, np.random.default_rng(n)), want[:1])
    assert emulator.opt_n_threads(n) == oracle.fps_block_size(n)
