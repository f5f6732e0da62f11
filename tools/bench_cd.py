"""Chamfer timing: exhaustive vs sorted kernel.  python tools/bench_cd.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvp_benchmark_amd import _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (b, n, m) in [(64, 16384, 16384), (32, 16384, 16384), (64, 8192, 8192), (64, 4096, 4096), (64, 2048, 16384), (64, 8192, 16384), (64, 2048, 2048)]:
    a = torch.rand(b, n, 3, generator=g).to(dev); c = torch.rand(b, m, 3, generator=g).to(dev)
    d1, d2 = torch.zeros(b, n, device=dev), torch.zeros(b, m, device=dev)
    i1, i2 = torch.zeros(b, n, dtype=torch.int32, device=dev), torch.zeros(b, m, dtype=torch.int32, device=dev)
    nbytes = _lib.chamfer_scratch_bytes(b, n, m); scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    res = []
    for name in ("mvp_chamfer_forward", "mvp_chamfer_forward_sorted"):
        args = (b, n, m, a, c, d1, d2, i1, i2) + ((scratch, nbytes) if name.endswith("sorted") else ())
        _lib.call(name, dev, *args); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); _lib.call(name, dev, *args); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        res.append(best)
    print("cd (%d,%d)x(%d): exhaustive %.3f ms, sorted %.3f ms (Q=%s)" % (b, n, m, res[0], res[1], os.environ.get("MVP_CD_Q", "default")), flush=True)
