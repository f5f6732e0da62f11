"""EMD timing sweep (BASELINE cfg 4: N in {1024..8192}, batch 64; plus headline).
   python tools/bench_emd.py [lib.so]   -- optional alternative library (profiling build)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvp_benchmark_amd import _lib
prof = len(sys.argv) > 1
if prof:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (b, n, eps, iters) in [(64, 1024, 0.004, 3000), (64, 2048, 0.004, 3000), (64, 4096, 0.004, 3000),
                           (64, 8192, 0.004, 3000), (64, 16384, 0.004, 3000), (64, 2048, 0.005, 50),
                           (64, 16384, 0.005, 50), (2, 2048, 0.004, 3000)]:
    x1 = torch.rand(b, n, 3, generator=g).to(dev); x2 = torch.rand(b, n, 3, generator=g).to(dev)
    nbytes = _lib.emd_scratch_bytes(b, n)
    scratch = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    dist = torch.zeros(b, n, device=dev); ass = torch.zeros(b, n, dtype=torch.int32, device=dev)
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call("mvp_emd_forward", dev, b, n, x1, x2, dist, ass, eps, iters, scratch, nbytes)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    st = scratch[nbytes - b * 16:].view(torch.int64).view(b, 2).cpu()
    if prof:   # the profiling build prints its per-phase cycle counters itself (clouds 0 and 1)
        print("b=%d n=%d eps=%g iters=%d: %.2f ms" % (b, n, eps, iters, best), flush=True)
    else:
        print("b=%d n=%d eps=%g iters=%d: %.2f ms  rounds %d bids/cloud %.0f  -> %.3g ref-pair-evals/s" % (
            b, n, eps, iters, best, int(st[:, 0].max()), st[:, 1].double().mean().item(),
            st[:, 1].double().sum().item() * n / (best * 1e-3)), flush=True)
