"""gather_max (edge_preserve_sampling's neighbour max-pool) at VRCNet's three pooling levels: forward and backward kernels.
MVP_LIB=<other libmvpops.so> for an A/B of two builds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvp_benchmark_amd import _lib
if os.environ.get("MVP_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MVP_LIB"])
from mvp_benchmark_amd.mm3d_pn2.functional import gather_max
dev = "cuda:0"
def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator().manual_seed(0)
for (B, C, N, P, k) in [(64, 64, 3072, 1536, 10), (64, 128, 1536, 768, 10), (64, 256, 768, 384, 10), (32, 120, 2048, 1024, 16)]:
    x = torch.randn(B, C, N, generator=g).to(dev).requires_grad_()
    idx = torch.randint(0, N, (B, P, k), generator=g, dtype=torch.int32).to(dev)
    out = gather_max(x, idx)
    go = torch.randn_like(out)
    fwd = timeit(lambda: gather_max(x.detach(), idx))
    bwd = timeit(lambda: torch.autograd.grad(out, x, go, retain_graph=True))
    print("gather_max (%d,%d,%d)->%d k=%d: forward %.1f us (%.0f GB/s of in + out), backward %.1f us" % (
        B, C, N, P, k, fwd, (x.numel() + 2 * out.numel()) * 4 / fwd / 1e3, bwd), flush=True)
