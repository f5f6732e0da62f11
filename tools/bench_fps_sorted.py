"""Plain vs Morton-sorted FPS kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvp_benchmark_amd import _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (b, n, m) in [(64, 16384, 2048), (64, 8192, 1024), (64, 6144, 1024), (32, 16384, 2048)]:
    x = torch.rand(b, n, 3, generator=g).to(dev)
    nbytes = _lib.fps_scratch_bytes(b, n); ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    res = []
    for name in ("mvp_furthest_point_sampling", "mvp_furthest_point_sampling_sorted"):
        temp = torch.full((b, n), 1e10, device=dev); idx = torch.zeros(b, m, dtype=torch.int32, device=dev)
        args = (b, n, m, x, temp, idx) + ((ws, nbytes) if name.endswith("sorted") else ())
        _lib.call(name, dev, *args); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            temp.fill_(1e10)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); _lib.call(name, dev, *args); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        res.append((best, idx.clone()))
    print("fps (%d,%d)->%d: plain %.3f ms, sorted %.3f ms, equal %s" % (b, n, m, res[0][0], res[1][0], bool(torch.equal(res[0][1], res[1][1]))), flush=True)
