"""How far does the auction's result depend on the GetMax race of the reference?

emd_cuda.cu:181-194 lets the LAST writer among the bidders within 1e-6 of an
object's maximal increment win: which bidder that is depends on the GPU's thread
schedule.  The oracle / HIP kernel take the highest qualifying bidder (the
reference executed sequentially); this script runs the other extreme (the lowest)
beside it on BASELINE cfg 4's sizes (+ the headline size) and prints how many
clouds differ at all, how many assignments differ in those clouds and how far
mean(sqrt(dist)) moves.  CPU only (the oracle).  Output committed as
profiles/r2_emd_schedule_sensitivity.txt.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402

oracle.build()
oracle.set_num_threads(os.cpu_count() or 1)
clouds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sizes = [int(a) for a in sys.argv[2:]] or [1024, 2048, 4096, 8192, 16384]
print("clouds per size: %d; eval setting eps=0.004 iters=3000 and training setting eps=0.005 iters=50" % clouds)
print("%6s %6s %5s | %14s %22s %24s" % ("n", "eps", "iters", "clouds differ", "assignments differ (max)", "rel. change of mean sqrt d"))
for n in sizes:
    for eps, iters in ((0.004, 3000), (0.005, 50)):
        rng = np.random.default_rng(n)
        x1 = rng.random((clouds, n, 3), dtype=np.float32)
        x2 = rng.random((clouds, n, 3), dtype=np.float32)
        t0 = time.time()
        d_hi, a_hi, _, _ = oracle.emd_forward_ex(x1, x2, eps, iters, getmax_lowest=False)
        d_lo, a_lo, _, _ = oracle.emd_forward_ex(x1, x2, eps, iters, getmax_lowest=True)
        diff = (a_hi != a_lo).mean(1)
        rel = np.abs(np.sqrt(d_lo).mean(1) / np.sqrt(d_hi).mean(1) - 1)
        print("%6d %6g %5d | %8d of %-3d %22.4f %24.2e   (%.0f s)" % (
            n, eps, iters, int((diff > 0).sum()), clouds, diff.max(), rel.max(), time.time() - t0), flush=True)
