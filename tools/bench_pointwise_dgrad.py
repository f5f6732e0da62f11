"""Data gradient of the few-channel 1x1 convolutions: mvp_pointwise_dgrad against the library's convolution_backward.
python tools/bench_pointwise_dgrad.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvp_benchmark_amd import _lib
dev = torch.device("cuda:0")
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for shape, cout in [((32, 48, 16, 1024), 24), ((32, 24, 16, 1024), 24), ((64, 24, 16, 3072), 24), ((64, 64, 1, 3072), 4), ((64, 64, 3072), 3),
                    ((64, 16, 1, 3072), 64), ((64, 4, 1, 3072), 64), ((64, 64, 16, 768), 64), ((32, 8, 3072), 128)]:
    if cout > 64 or shape[1] > 64: continue
    x = torch.randn(*shape, device=dev); nd = len(shape) - 2
    w = torch.randn(cout, shape[1], *([1] * nd), device=dev)
    gy = torch.randn(shape[0], cout, *shape[2:], device=dev)
    gx = torch.empty_like(x)
    L = x[0, 0].numel()
    t_lib = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1] * nd, [0] * nd, [1] * nd, False, [0] * nd, 1, [True, False, False]))
    t_own = timeit(lambda: _lib.call("mvp_pointwise_dgrad", dev, shape[0], shape[1], cout, L, w, gy, gx))
    ref = torch.ops.aten.convolution_backward(gy, x, w, None, [1] * nd, [0] * nd, [1] * nd, False, [0] * nd, 1, [True, False, False])[0]
    err = float((gx - ref).abs().max() / ref.abs().max())
    gb = (gy.numel() + gx.numel()) * 4 / 1e9
    print("%-22s -> %2d: library %.3f ms, mvp_pointwise_dgrad %.3f ms (%.0f GB/s of gy + gx), rel err %.1e" % (shape, cout, t_lib, t_own, gb / t_own * 1e3, err))
