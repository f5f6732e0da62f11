"""Digest of the EMD results of one library build on fixed seeded inputs:
   python tools/emd_variant_hash.py [lib.so]
Variants built with `make -C mvp_benchmark_amd/csrc variant ...` must print the digests of the default
library (which the -m gpu tests pin to the oracle bit for bit)."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mvp_benchmark_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
dev = torch.device("cuda:0")
cases = []
rng = np.random.default_rng(5)
cases.append(("uniform 4x4096", rng.random((4, 4096, 3), dtype=np.float32), rng.random((4, 4096, 3), dtype=np.float32), 0.004, 3000))
cases.append(("dups 2x2048", np.tile(rng.random((2, 512, 3), dtype=np.float32), (1, 4, 1)), np.tile(rng.random((2, 256, 3), dtype=np.float32), (1, 8, 1)), 0.005, 1500))
cases.append(("blob 2x2048", (0.5 + 0.01 * rng.random((2, 2048, 3), dtype=np.float32)).astype(np.float32), rng.random((2, 2048, 3), dtype=np.float32), 0.004, 1500))
cases.append(("forced 2x2048", rng.random((2, 2048, 3), dtype=np.float32), rng.random((2, 2048, 3), dtype=np.float32), 0.002, 400))
g = torch.Generator().manual_seed(0)
cases.append(("headline 64x16384", torch.rand(64, 16384, 3, generator=g).numpy(), torch.rand(64, 16384, 3, generator=g).numpy(), 0.004, 3000))
for name, a, b_, eps, iters in cases:
    b, n = a.shape[:2]
    x1, x2 = torch.from_numpy(a).to(dev), torch.from_numpy(b_).to(dev)
    nbytes = _lib.emd_scratch_bytes(b, n)
    scratch = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    dist = torch.zeros(b, n, device=dev); ass = torch.zeros(b, n, dtype=torch.int32, device=dev)
    _lib.call("mvp_emd_forward", dev, b, n, x1, x2, dist, ass, eps, iters, scratch, nbytes)
    torch.cuda.synchronize()
    st = scratch[nbytes - b * 16:].view(torch.int64).view(b, 2).cpu()
    h = hashlib.sha1(dist.cpu().numpy().tobytes() + ass.cpu().numpy().tobytes()).hexdigest()[:16]
    print("%-20s %s rounds %d bids %d" % (name, h, int(st[:, 0].max()), int(st[:, 1].sum())), flush=True)
