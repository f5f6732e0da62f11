"""Property-based CPU checks of the oracle against brute-force NumPy
definitions on random shapes (hypothesis)."""
import numpy as np
from hypothesis import given, settings, strategies as st


def _clouds(seed, b, n):
    return np.random.default_rng(seed).random((b, n, 3), dtype=np.float32)


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 10_000), st.integers(1, 3), st.integers(1, 90), st.integers(1, 90))
def test_chamfer_equals_bruteforce(oracle, seed, b, n, m):
    a, c = _clouds(seed, b, n), _clouds(seed + 1, b, m)
    d1, d2, i1, i2 = oracle.chamfer_forward(a, c)
    full = ((a[:, :, None].astype(np.float64) - c[:, None].astype(np.float64)) ** 2).sum(-1)
    np.testing.assert_allclose(d1, full.min(2), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(d2, full.min(1), rtol=1e-5, atol=1e-9)
    # the returned index attains the minimum (ties may pick any minimiser in float64 terms)
    np.testing.assert_allclose(np.take_along_axis(full, i1[..., None].astype(np.int64), 2)[..., 0], full.min(2),
                               rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(np.take_along_axis(full.transpose(0, 2, 1), i2[..., None].astype(np.int64), 2)[..., 0],
                               full.min(1), rtol=1e-5, atol=1e-9)


@settings(max_examples=15, deadline=None)
@given(st.integers(0, 10_000), st.integers(2, 200), st.integers(1, 40))
def test_fps_indices_are_valid_and_greedy(oracle, seed, n, m):
    x = _clouds(seed, 1, n)
    idx = oracle.furthest_point_sample(x, m)[0]
    assert idx[0] == 0 and idx.min() >= 0 and idx.max() < n
    temp = np.full(n, np.inf)
    for j in range(1, m):
        temp = np.minimum(temp, ((x[0].astype(np.float64) - x[0, idx[j - 1]]) ** 2).sum(-1))
        assert temp[idx[j]] >= temp.max() * (1 - 1e-5)


@settings(max_examples=15, deadline=None)
@given(st.integers(0, 10_000), st.integers(1, 8), st.integers(3, 120), st.integers(1, 60))
def test_knn_and_three_nn_sets(oracle, seed, k, n, m):
    k = min(k, n)
    xyz, ctr = _clouds(seed, 1, n), _clouds(seed + 1, 1, m)
    idx, d = oracle.knn(k, xyz, ctr, return_dist=True)
    full = ((ctr[:, :, None].astype(np.float64) - xyz[:, None].astype(np.float64)) ** 2).sum(-1)
    np.testing.assert_allclose(d, np.sort(full, -1)[..., :k], rtol=1e-5, atol=1e-9)
    dist, i3 = oracle.three_nn(ctr, xyz)
    np.testing.assert_allclose(dist ** 2, np.sort(full, -1)[..., :3], rtol=1e-4, atol=1e-8)
