"""Randomised consistency campaign of the tiered EMD launch: for random batch sizes (33..64), cloud sizes and input
distributions the default (split = 2) must give the bits of the first kernel alone (split = 0) and the same statistics.
python tools/fuzz_emd_tiers.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mvp_benchmark_amd import _lib
dev = torch.device("cuda:0")
cases, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
def run(x1, x2, eps, iters, split):
    _lib.emd_configure(split=split)
    b, n = x1.shape[:2]
    nbytes = _lib.emd_scratch_bytes(b, n); scratch = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    dist = torch.zeros(b, n, device=dev); ass = torch.zeros(b, n, dtype=torch.int32, device=dev)
    _lib.call("mvp_emd_forward", dev, b, n, x1, x2, dist, ass, eps, iters, scratch, nbytes); torch.cuda.synchronize()
    stats = scratch[nbytes - b * 16:].view(torch.int64).view(b, 2).cpu()
    return dist.cpu(), ass.cpu(), stats, _lib.emd_records(scratch, nbytes, b)
bad = 0
try:
    for c in range(cases):
        b = int(rng.integers(33, 65)); n = int(rng.choice([4096, 4096, 8192])); iters = int(rng.choice([700, 1200, 3000])); eps = float(rng.choice([0.004, 0.002, 0.008]))
        kind = rng.choice(["uniform", "mixed", "shells", "near", "dups"])
        g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
        x2 = torch.rand(b, n, 3, generator=g)
        if kind == "uniform": x1 = torch.rand(b, n, 3, generator=g)
        elif kind == "mixed":
            x1 = torch.rand(b, n, 3, generator=g); x1[::3] = (0.5 + 0.2 * torch.randn(len(x1[::3]), n, 3, generator=g)).clamp(0, 1)
        elif kind == "shells":
            s = torch.randn(b, n, 3, generator=g); x1 = 0.5 + 0.45 * s / s.norm(dim=2, keepdim=True)
        elif kind == "near": x1 = (x2 + 0.02 * torch.randn(b, n, 3, generator=g)).clamp(0, 1)
        else:
            x1 = torch.rand(b, n // 4, 3, generator=g).repeat(1, 4, 1); x2 = torch.rand(b, n // 2, 3, generator=g).repeat(1, 2, 1)
        x1, x2 = x1.to(dev).contiguous(), x2.to(dev).contiguous()
        d0, a0, s0, _ = run(x1, x2, eps, iters, 0)
        d2, a2, s2, rec = run(x1, x2, eps, iters, 2)
        ok = torch.equal(d0, d2) and torch.equal(a0, a2) and torch.equal(s0, s2)
        tiered = rec["final_launch"] == 2
        widths = sorted(set(rec["final_width"][tiered].tolist()))
        print("case %2d: b %2d n %5d iters %4d eps %.3f %-7s -> %s; %2d clouds finished by the tiered launch, widths %s" % (
            c, b, n, iters, eps, kind, "identical" if ok else "MISMATCH", int(tiered.sum()), widths), flush=True)
        bad += 0 if ok else 1
finally:
    _lib.emd_configure(split=_lib.EMD_DEFAULT_SPLIT)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
