"""Clouds of a batch one at a time: python tools/emd_per_cloud.py lib.so N [K]  -- the first K (16) clouds of bench_emd_one.py's
(64, N) batch, each in a call of its own (best of 5), with its rounds and bids: what a change costs or saves PER CLOUD, which the
batch's time (the slowest cloud's) hides."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvp_benchmark_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
n = int(sys.argv[2]); k = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x1 = torch.rand(64, n, 3, generator=g).to(dev); x2 = torch.rand(64, n, 3, generator=g).to(dev)
out = []
for c in range(k):
    a, b_ = x1[c:c + 1].contiguous(), x2[c:c + 1].contiguous()
    nbytes = _lib.emd_scratch_bytes(1, n)
    scratch = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    dist = torch.zeros(1, n, device=dev); ass = torch.zeros(1, n, dtype=torch.int32, device=dev)
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _lib.call("mvp_emd_forward", dev, 1, n, a, b_, dist, ass, 0.004, 3000, scratch, nbytes); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    st = scratch[nbytes - 16:].view(torch.int64).cpu()
    out.append("%.2f(%d/%d)" % (best, int(st[0]), int(st[1])))
print(os.path.basename(sys.argv[1]), "n=%d ms(rounds/bids):" % n, " ".join(out), flush=True)
