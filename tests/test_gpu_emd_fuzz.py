"""A fixed-seed slice of tools/fuzz_emd_tiers.py under pytest -m gpu (the full campaign's log is
profiles/r4_fuzz_emd.txt): random batch sizes, cloud sizes, settings and input distributions -- every launch sequence of
mvp_emd_forward (first kernel alone / + lean + tiered widths / + LDS-resident tail) gives the same bits and statistics
(utils/metrics/EMD/emd_cuda.cu:95-226; the first kernel alone is pinned to the oracle in test_gpu_ops.py)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def _cases(count, seed):
    import fuzz_emd_tiers as fz
    rng = np.random.default_rng(seed)
    return [fz.draw_case(rng) for _ in range(count)]


@pytest.mark.parametrize("case", _cases(12, 2025), ids=lambda c: "b%d-n%d-it%d-eps%g-%s-w%d" % (c[:5] + (c[6],)))
def test_emd_launch_sequences_agree_on_random_cases(case):
    import fuzz_emd_tiers as fz
    ok, what = fz.run_case(case)
    assert ok, (case, what)


def test_fuzz_slice_covers_every_family():
    """The slice holds tiered batches (33..64 clouds of >= 4096 points), resident-size clouds, clusters of ONE workgroup
    (forced, and by batch size: > 128 clouds) and several input kinds."""
    cases = _cases(12, 2025)
    assert any(33 <= c[0] <= 64 and c[1] >= 4096 for c in cases) and any(c[1] <= 2048 for c in cases)
    assert any(c[6] == 1 for c in cases) and any(c[0] > 128 for c in cases)
    assert len(set(c[4] for c in cases)) >= 3
