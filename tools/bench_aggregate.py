"""Shared-weight neighbourhood aggregation (SA_module): fused op against the PyTorch formulation, fwd and fwd+bwd."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvp_benchmark_amd.mm3d_pn2.functional import share_weighted_sum
dev = "cuda:0"

def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for (B, share, Cw, k, N) in [(64, 8, 2, 16, 3072), (64, 8, 4, 16, 1536), (64, 8, 8, 16, 768), (64, 8, 16, 16, 384)]:
    w = torch.randn(B, Cw, k, N, device=dev, requires_grad=True)
    v = torch.randn(B, share * Cw, k, N, device=dev, requires_grad=True)
    go = torch.randn(B, share * Cw, N, device=dev)
    f_op = lambda: share_weighted_sum(w, v)
    f_th = lambda: (w.unsqueeze(1) * v.view(B, share, Cw, k, N)).sum(dim=3)
    def fb(f):
        return lambda: torch.autograd.grad(f(), (w, v), go.view_as(f()))
    mb = 4.0 * v.numel() / 1e6
    print("(%d,%dx%d,%d,%d) v = %.0f MB: fwd op %.3f ms torch %.3f ms | fwd+bwd op %.3f ms torch %.3f ms" % (
        B, share, Cw, k, N, mb, timeit(f_op), timeit(f_th), timeit(fb(f_op)), timeit(fb(lambda: f_th().reshape(B, share * Cw, N)))), flush=True)
