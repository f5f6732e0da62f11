"""Compare mvp_emd_forward against the oracle for one cluster width (MVP_EMD_CLUSTER) on a few cases."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from mvp_benchmark_amd.metrics import emd
oracle.build()
dev = torch.device("cuda")
def rc(seed, b, n): return np.random.default_rng(seed).random((b, n, 3), dtype=np.float32)
cases = []
for iters in (1, 2, 3, 5, 10, 50, 300):
    cases.append(("cluster it=%d" % iters, (0.5 + 0.01 * rc(0, 2, 2048)).astype(np.float32), rc(1, 2, 2048), 0.004, iters))
cases.append(("uniform", rc(2, 3, 4096), rc(3, 3, 4096), 0.004, 3000))
for name, x1, x2, eps, iters in cases:
    d, a = emd()(torch.from_numpy(x1).to(dev), torch.from_numpy(x2).to(dev), eps, iters)
    od, oa = oracle.emd_forward(x1, x2, eps, iters)
    a = a.cpu().numpy(); d = d.cpu().numpy()
    print(name, "W=%s" % os.environ.get("MVP_EMD_CLUSTER"), "ass mismatches", int((a != oa).sum()), "dist mismatches", int((d != od).sum()))
