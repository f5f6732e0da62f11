import sys, os, torch
sys.path.insert(0, os.getcwd())
from mvp_benchmark_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
only_sorted = len(sys.argv) > 1
for (b, n, k) in [(64, 16384, 16), (64, 8192, 16), (64, 4096, 16), (64, 16384, 8), (64, 16384, 32)]:
    x = torch.rand(b, n, 3, generator=g).to(dev)
    for name, fn in (("sorted", "mvp_knn_sorted"),) + (() if only_sorted else (("exhaustive", "mvp_knn"),)):
        nbytes = _lib.knn_scratch_bytes(b, n, n)
        scratch = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        idx = torch.zeros(b, n, k, dtype=torch.int32, device=dev); d2 = torch.zeros(b, n, k, device=dev)
        def run():
            if fn == "mvp_knn_sorted":
                _lib.call(fn, dev, b, n, n, k, x, x, idx, d2, scratch, nbytes)
            else:
                _lib.call(fn, dev, b, n, n, k, x, x, idx, d2)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): run()
        e1.record(); torch.cuda.synchronize()
        print("knn k=%d (%d,%d) %s: %.3f ms" % (k, b, n, name, e0.elapsed_time(e1) / 3), flush=True)
