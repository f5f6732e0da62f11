"""EMD on the degenerate input of BASELINE cfg 2 with random-init weights: the
prediction is one tight blob, the target is spread (every bid scans every object)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvp_benchmark_amd import _lib
b, n = int(sys.argv[1]), int(sys.argv[2]); iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x1 = (0.5 + 0.01 * torch.rand(b, n, 3, generator=g)).to(dev); x2 = torch.rand(b, n, 3, generator=g).to(dev)
nbytes = _lib.emd_scratch_bytes(b, n); scratch = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
dist = torch.zeros(b, n, device=dev); ass = torch.zeros(b, n, dtype=torch.int32, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); _lib.call("mvp_emd_forward", dev, b, n, x1, x2, dist, ass, 0.004, iters, scratch, nbytes); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
st = scratch[nbytes - b * 16:].view(torch.int64).view(b, 2).cpu()
bids = st[:, 1].double().sum().item()
print("clustered b=%d n=%d iters=%d: %.1f ms, rounds %d, bids/cloud %.0f -> %.3g object scans/s" % (b, n, iters, ms, int(st[:, 0].max()), bids / b, bids * n / (ms * 1e-3)))
