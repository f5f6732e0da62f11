"""Randomised consistency campaign of the EMD launch sequences: for random batch sizes, cloud sizes, settings and input
distributions every `split` (0: the first kernel alone; 2: + lean kernel + widths dealt out by load at round 300;
3: + LDS-resident tail for clouds of <= 4096 points in a launch of its own; 4: fused into the lean launch;
5, the default: + gathered-bid rounds once at most 256 persons of a cloud are unassigned) must give the same bits and the same statistics.

  python tools/fuzz_emd_tiers.py [cases] [seed] [few]    (few: the cases of draw_case_few; log of the round's campaign: profiles/r4_fuzz_emd.txt)

tests/test_gpu_emd_fuzz.py runs a fixed-seed slice of the same cases under pytest -m gpu."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

KINDS = ["uniform", "mixed", "shells", "near", "dups", "surface"]


def draw_case(rng):
    """One random case: (b, n, iters, eps, kind, input seed, cluster).  Three families: the batches the tiered launch
    serves (33..64 clouds of 4096 / 8192 points), the small clouds the resident tail serves (any batch), and -- round 5:
    the round-4 campaign never drew one and missed a W = 1 bug -- clusters of ONE workgroup: batches of more than 128
    clouds (the automatic width) or `cluster` forced to 1 / 2 on small batches."""
    r = rng.random()
    cluster = 0
    if r < 0.4:
        b, n = int(rng.integers(33, 65)), int(rng.choice([4096, 4096, 8192]))
    elif r < 0.75:
        b, n = int(rng.choice([1, 2, 3, 7, 16, 40, 64])), int(rng.choice([1024, 2048, 2048, 3072, 4096]))
    elif r < 0.85:
        b, n = int(rng.choice([129, 160, 200])), int(rng.choice([1024, 2048]))
    else:
        b, n, cluster = int(rng.choice([1, 3, 16, 40])), int(rng.choice([1024, 2048, 4096, 8192])), int(rng.choice([1, 1, 2]))
    iters = int(rng.choice([700, 1200, 3000]))
    eps = float(rng.choice([0.004, 0.002, 0.008]))
    return b, n, iters, eps, str(rng.choice(KINDS)), int(rng.integers(1 << 30)), cluster


def draw_case_few(rng):
    """Round 6: cases for the rounds of at most 16 bidders (emd_lean_round_few.inc) -- clouds above the resident tail's
    4096 points whose auction thins out: a prediction near its ground truth (noise 0.02 .. 0.1 of the unit cube), small and
    full batches, every cluster width, short and long auctions.  split 0 (the first kernel alone) never runs those rounds:
    it is the reference the others are compared with."""
    b = int(rng.choice([1, 2, 3, 5, 8, 33, 48, 64]))
    n = int(rng.choice([5120, 6144, 8192, 8192, 12288, 16384]))
    if b > 8 and n > 8192:
        b = int(rng.choice([4, 6]))   # (keeps the campaign's run time down)
    cluster = int(rng.choice([0, 0, 0, 1, 2, 4, 8])) if b <= 8 else 0
    iters = int(rng.choice([400, 700, 1200, 3000]))
    eps = float(rng.choice([0.004, 0.002, 0.008]))
    return b, n, iters, eps, str(rng.choice(["near", "near2", "near5", "neardup"])), int(rng.integers(1 << 30)), cluster


def make_inputs(b, n, kind, seed):
    g = torch.Generator().manual_seed(seed)
    x2 = torch.rand(b, n, 3, generator=g)
    if kind == "uniform":
        x1 = torch.rand(b, n, 3, generator=g)
    elif kind == "mixed":
        x1 = torch.rand(b, n, 3, generator=g)
        x1[::3] = (0.5 + 0.2 * torch.randn(len(x1[::3]), n, 3, generator=g)).clamp(0, 1)
    elif kind == "shells":
        s = torch.randn(b, n, 3, generator=g)
        x1 = 0.5 + 0.45 * s / s.norm(dim=2, keepdim=True)
    elif kind in ("near", "near2", "near5"):
        x1 = (x2 + {"near": 0.02, "near2": 0.05, "near5": 0.1}[kind] * torch.randn(b, n, 3, generator=g)).clamp(0, 1)
    elif kind == "neardup":   # duplicated points, near pairs: ties and contests while few persons bid
        x2 = torch.rand(b, n // 4, 3, generator=g).repeat(1, 4, 1)
        x1 = (x2 + 0.08 * (torch.rand(b, n // 4, 3, generator=g) - 0.5).repeat(1, 4, 1)).clamp(0, 1)
    elif kind == "dups":
        x1 = torch.rand(b, n // 4, 3, generator=g).repeat(1, 4, 1)
        x2 = torch.rand(b, n // 2, 3, generator=g).repeat(1, 2, 1)
    else:   # both clouds on one torus: dense cells (hundreds of objects in an occupied cell)
        def torus():
            u, v = 2 * np.pi * torch.rand(b, n, generator=g), 2 * np.pi * torch.rand(b, n, generator=g)
            return torch.stack([0.5 + (0.3 + 0.1 * torch.cos(v)) * torch.cos(u), 0.5 + (0.3 + 0.1 * torch.cos(v)) * torch.sin(u),
                                0.5 + 0.1 * torch.sin(v)], 2)
        x1, x2 = torus(), torus()
    return x1.float().contiguous(), x2.float().contiguous()


def run(x1, x2, eps, iters, split, dev, cluster=0):
    from mvp_benchmark_amd import _lib
    _lib.emd_configure(split=split, cluster=cluster)
    b, n = x1.shape[:2]
    nbytes = _lib.emd_scratch_bytes(b, n)
    scratch = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    dist = torch.zeros(b, n, device=dev)
    ass = torch.zeros(b, n, dtype=torch.int32, device=dev)
    _lib.call("mvp_emd_forward", dev, b, n, x1, x2, dist, ass, eps, iters, scratch, nbytes)
    torch.cuda.synchronize()
    stats = scratch[nbytes - b * 16:].view(torch.int64).view(b, 2).cpu()
    return dist.cpu(), ass.cpu(), stats, _lib.emd_records(scratch, nbytes, b)


def run_case(case, dev="cuda:0"):
    """-> (identical, description of what ran)."""
    from mvp_benchmark_amd import _lib
    b, n, iters, eps, kind, seed, cluster = case
    x1, x2 = make_inputs(b, n, kind, seed)
    x1, x2 = x1.to(dev), x2.to(dev)
    try:
        d0, a0, s0, _ = run(x1, x2, eps, iters, 0, dev, cluster)
        ok, notes = True, []
        for split in (2, 3, 4, 5):
            d, a, s, rec = run(x1, x2, eps, iters, split, dev, cluster)
            ok = ok and torch.equal(d0, d) and torch.equal(a0, a) and torch.equal(s0, s)
            fl = rec["final_launch"]
            notes.append("split %d: launches %s widths %s" % (split, sorted(set(fl.tolist())),
                                                             sorted(set(rec["final_width"][fl > 0].tolist()))))
    finally:
        _lib.emd_configure(split=_lib.EMD_DEFAULT_SPLIT, cluster=0)
    return ok, "; ".join(notes)


def main():
    cases, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(cases):
        case = draw_case_few(rng) if len(sys.argv) > 3 and sys.argv[3] == "few" else draw_case(rng)
        ok, what = run_case(case)
        print("case %2d: b %3d n %5d iters %4d eps %.3f %-7s cluster %d -> %s | %s" % (
            (c,) + case[:5] + (case[6], "identical" if ok else "MISMATCH", what)), flush=True)
        bad += 0 if ok else 1
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
