"""GPU debug harness for the EMD kernel: tiny cases, per-case watchdog."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from mvp_benchmark_amd import _lib

def run(b, n, eps, iters, seed=0):
    rng = np.random.default_rng(seed)
    x1 = rng.random((b, n, 3), dtype=np.float32); x2 = rng.random((b, n, 3), dtype=np.float32)
    dev = torch.device("cuda:0")
    t1, t2 = torch.tensor(x1, device=dev), torch.tensor(x2, device=dev)
    nbytes = _lib.emd_scratch_bytes(b, n)
    scratch = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    dist = torch.zeros(b, n, device=dev); ass = torch.zeros(b, n, dtype=torch.int32, device=dev)
    t0 = time.time()
    _lib.call("mvp_emd_forward", dev, b, n, t1, t2, dist, ass, eps, iters, scratch, nbytes)
    torch.cuda.synchronize()
    dt = time.time() - t0
    stats = scratch[nbytes - b * 16:].view(torch.int64).view(b, 2).cpu().tolist()
    od, oa, ost = oracle.emd_forward(x1, x2, eps, iters, return_stats=True)
    ok = np.array_equal(ass.cpu().numpy(), oa) and np.array_equal(dist.cpu().numpy(), od)
    print("b=%d n=%d eps=%g iters=%d: %.1f ms gpu stats %s oracle stats %s parity %s" % (
        b, n, eps, iters, dt * 1e3, stats, ost.tolist(), ok), flush=True)
    return ok

if __name__ == "__main__":
    cases = [(1, 1024, 0.005, 1), (1, 1024, 0.005, 2), (1, 1024, 0.005, 50), (2, 2048, 0.004, 3000),
             (2, 3072, 0.005, 50), (4, 8192, 0.004, 3000), (64, 16384, 0.004, 3000)]
    if len(sys.argv) > 1:
        cases = cases[:int(sys.argv[1])]
    for c in cases:
        run(*c)
