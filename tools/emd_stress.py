"""Randomised EMD parity stress: many small shapes / settings / cluster widths against the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from mvp_benchmark_amd.metrics import emd
from mvp_benchmark_amd import _lib
oracle.build()
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
t0 = time.time()
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    b = int(rng.integers(1, 6)); n = int(rng.choice([1024, 2048, 3072, 4096])); eps = float(rng.choice([0.002, 0.004, 0.005, 0.02]))
    iters = int(rng.choice([1, 2, 7, 50, 300, 1500])); kind = int(rng.integers(0, 4))
    x1 = rng.random((b, n, 3), dtype=np.float32); x2 = rng.random((b, n, 3), dtype=np.float32)
    if kind == 1: x1 = (0.5 + 0.02 * x1).astype(np.float32)                      # clustered prediction
    if kind == 2: x2 = np.round(x2 * 8).astype(np.float32) / 8                    # lattice target: many ties
    if kind == 3: x1[:, n // 2:] = x1[:, :n // 2]                                  # duplicated persons
    od, oa = oracle.emd_forward(x1, x2, eps, iters)
    for w in (1, 2, 4, 8):
        _lib.emd_configure(cluster=w)
        d, a = emd()(torch.from_numpy(x1).to(dev), torch.from_numpy(x2).to(dev), eps, iters)
        ok = np.array_equal(a.cpu().numpy(), oa) and np.array_equal(d.cpu().numpy(), od)
        if not ok:
            bad += 1
            print("MISMATCH case %d: b=%d n=%d eps=%g iters=%d kind=%d W=%d" % (case, b, n, eps, iters, kind, w), flush=True)
print("stress: %d mismatching (case, width) pairs, %.0f s" % (bad, time.time() - t0))
sys.exit(1 if bad else 0)
