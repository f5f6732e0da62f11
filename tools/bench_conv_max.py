"""mvp_pointwise_max_backward (sparse backward of conv -> max over the positions) against the dense route
(max's scatter + MFMA data / weight gradient) at the networks' shapes.  python tools/bench_conv_max.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvp_benchmark_amd import _lib
from mvp_benchmark_amd.pointwise import mfma_linear, mfma_wgrad
dev = "cuda:0"

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for (B, cin, cout, L) in [(64, 512, 1024, 2048), (64, 512, 1024, 384), (32, 1800, 1024, 64), (32, 512, 1024, 2048)]:
    x = torch.randn(B, cin, L, device=dev); w = torch.randn(cout, cin, device=dev)
    y = mfma_linear(x, w); val, idx = y.max(dim=2); idx32 = idx.int()
    g = torch.randn(B, cout, device=dev)
    gx = torch.empty_like(x); gw = torch.empty_like(w); gb = torch.empty(cout, device=dev)
    t_dg = timeit(lambda: _lib.call("mvp_pointwise_max_backward", dev, B, cin, cout, L, x, w, g, idx32, gx, None, None, None, 0))
    t_wg0 = timeit(lambda: _lib.call("mvp_pointwise_max_backward", dev, B, cin, cout, L, x, w, g, idx32, None, gw, gb, None, 0))
    nb = _lib.pointwise_max_backward_scratch_bytes(B, cin, cout, L); ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
    t_wg = timeit(lambda: _lib.call("mvp_pointwise_max_backward", dev, B, cin, cout, L, x, w, g, idx32, None, gw, gb, ws, nb))
    def dense():
        gy = torch.zeros_like(y).scatter_(2, idx.unsqueeze(2), g.unsqueeze(2))
        mfma_linear(gy, w, w_kmajor=True); mfma_wgrad(x, gy, cout, cin, True)
    t_dense = timeit(dense)
    print("(%d,%d->%d,%d): sparse dgrad %.3f ms, sparse wgrad %.3f ms (direct gather %.3f); dense route %.3f ms" % (B, cin, cout, L, t_dg, t_wg, t_wg0, t_dense), flush=True)
