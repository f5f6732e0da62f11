import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from mvp_benchmark_amd.metrics import emd
oracle.build()
dev = torch.device("cuda")
def rc(seed, b, n): return np.random.default_rng(seed).random((b, n, 3), dtype=np.float32)
x1 = (0.5 + 0.01 * rc(0, 2, 2048)).astype(np.float32); x2 = rc(1, 2, 2048)
for iters in range(10, 40):
    d, a = emd()(torch.from_numpy(x1).to(dev), torch.from_numpy(x2).to(dev), 0.004, iters)
    od, oa = oracle.emd_forward(x1, x2, 0.004, iters)
    a = a.cpu().numpy()
    bad = np.argwhere(a != oa)
    print(iters, "mismatches", len(bad), bad[:4].tolist(), [(int(a[i, j]), int(oa[i, j])) for i, j in bad[:4]])
    if len(bad): break
