"""Runs tests/test_model_golden.py and prints, for every comparison with the reference-generated fixtures, the error
actually measured next to the tolerance, and for every replayed pass the index rows that differed from the reference
run (all of them verified ties).  Usage: python tests/report_model_golden.py [gpu]   (lives under tests/: it drives tests that use the CPU oracle)  ->  stdout (profiles/r5_model_golden_*.txt)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_model_golden as tm  # noqa: E402

_close, _close_l2, _flips = tm.close, tm.close_l2, tm._check_flips


def _np(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def close(got, want, rel, what=""):
    w = np.asarray(want)
    e = float(np.abs(_np(got).astype(np.float64) - w).max()) if w.size else 0.0
    print("  max|err|/max|want|  %-42s %.2e  (tolerance %.0e)" % (what, e / max(float(np.abs(w).max()) if w.size else 0, 1e-30), rel))
    _close(got, want, rel, what)


def close_l2(got, want, rel, what=""):
    w = np.asarray(want)
    print("  ||err||/||want||    %-42s %.2e  (tolerance %.0e)" % (what, np.linalg.norm(_np(got) - w) / max(np.linalg.norm(w), 1e-30), rel))
    _close_l2(got, want, rel, what)


def check_flips(rp, budget):
    bad = {k: v for k, v in rp.flips.items() if v[0]}
    print("  index rows differing from the reference run: %d of %d %s; unexplained: %d"
          % (rp.total_flips()[0], rp.total_flips()[1], bad or "", len(rp.unexplained)))
    _flips(rp, budget)


tm.close, tm.close_l2, tm._check_flips = close, close_l2, check_flips
gpu = len(sys.argv) > 1 and sys.argv[1] == "gpu"
sys.exit(pytest.main(["-q", "-s", "-m", "gpu" if gpu else "not gpu", os.path.join(ROOT, "tests", "test_model_golden.py"),
                      "-p", "no:cacheprovider"]))
