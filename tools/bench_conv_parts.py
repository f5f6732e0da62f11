"""MIOpen 1x1 convolution split into forward / data gradient / weight gradient for the small-channel per-edge and
per-point layers of ECG and VRCNet, with the bytes each pass has to move."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from mvp_benchmark_amd import _lib
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

for (shp, cout) in [((32, 24, 16, 3072), 24), ((32, 48, 16, 3072), 24), ((32, 48, 16, 1024), 24), ((64, 256, 1, 768), 16),
                    ((64, 64, 1, 3072), 4), ((64, 16, 1, 3072), 64), ((64, 128, 1, 1536), 32), ((64, 512, 1, 384), 32)]:
    b, cin = shp[0], shp[1]
    L = shp[2] * shp[3]
    x = torch.randn(*shp, device=dev)
    w = torch.randn(cout, cin, 1, 1, device=dev)
    bias = torch.randn(cout, device=dev)
    y = F.conv2d(x, w, bias)
    gy = torch.randn_like(y)
    bw = lambda mask: torch.ops.aten.convolution_backward(gy, x, w, [cout], [1, 1], [0, 0], [1, 1], False, [0, 0], 1, mask)
    tf = timeit(lambda: F.conv2d(x, w, bias))
    td = timeit(lambda: bw([True, False, False]))
    tw = timeit(lambda: bw([False, True, True]))
    x3, gy3 = x.flatten(2), gy.flatten(2)
    tb = timeit(lambda: (torch.bmm(gy3, x3.transpose(1, 2)).sum(0), gy3.sum((0, 2))))      # weight gradient as a batched GEMM
    nbytes = _lib.pointwise_wgrad_scratch_bytes(b, cin, cout, L)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev); gw = torch.empty_like(w); gb = torch.empty(cout, device=dev)
    tk = timeit(lambda: _lib.call("mvp_pointwise_wgrad", dev, b, cin, cout, L, x, gy, gw, gb, ws, nbytes))
    mb = lambda nch: 4.0 * b * L * nch / 1e6
    print("%-22s -> %3d: fwd %.3f ms (%.0f MB: %.2f TB/s)  dgrad %.3f ms (%.2f TB/s)  wgrad %.3f ms (%.2f TB/s)  wgrad as bmm %.3f ms  mvp_pointwise_wgrad %.3f ms (%.2f TB/s)" % (
        shp, cout, tf, mb(cin + cout), mb(cin + cout) / tf / 1e3, td, mb(cin + cout) / td / 1e3, tw, mb(cin + cout) / tw / 1e3, tb, tk, mb(cin + cout) / tk / 1e3), flush=True)
