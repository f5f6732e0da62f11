"""mvp_pointwise_mfma against the library convolution on the >= 32-channel 1x1-convolution
shapes of PCN / VRCNet (forward; data gradient = the same kernel with the transposed weight)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from mvp_benchmark_amd import _lib
if os.environ.get('MVP_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['MVP_LIB'])
from mvp_benchmark_amd.pointwise import mfma_linear
dev = "cuda:0"

def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

print("%-28s %10s %10s %8s | %10s %10s" % ("shape (B,Cin->Cout,L)", "mfma ms", "lib ms", "speedup", "mfma TF/s", "lib TF/s"))
for (B, cin, cout, L) in [(64, 128, 256, 2048), (64, 512, 512, 2048), (64, 512, 1024, 2048), (64, 1090, 256, 2048),
                          (64, 64, 128, 3072), (64, 128, 128, 1536), (64, 256, 256, 768), (64, 512, 512, 384),
                          (64, 256, 64, 3072), (64, 64, 64, 3072), (32, 128, 256, 16384), (64, 1024, 512, 384)]:
    x = torch.randn(B, cin, L, device=dev); w = torch.randn(cout, cin, device=dev); b = torch.randn(cout, device=dev)
    w3 = w.unsqueeze(2).contiguous()
    t1 = timeit(lambda: mfma_linear(x, w, b, relu=True))
    t2 = timeit(lambda: torch.relu_(F.conv1d(x, w3, b)))
    fl = 2.0 * B * cin * cout * L
    print("%-28s %10.3f %10.3f %8.2f | %10.1f %10.1f" % ("(%d,%d->%d,%d)" % (B, cin, cout, L), t1, t2, t2 / t1, fl / t1 / 1e9, fl / t2 / 1e9), flush=True)

from mvp_benchmark_amd.pointwise import mfma_wgrad
print()
print("%-28s %10s %10s %8s | %10s %10s   (data gradient W^T g / weight+bias gradient)" % ("shape (B,Cin->Cout,L)", "mfma ms", "lib ms", "speedup", "mfma TF/s", "lib TF/s"))
for (B, cin, cout, L) in [(64, 128, 256, 2048), (64, 512, 512, 2048), (64, 512, 1024, 2048), (64, 1090, 256, 2048),
                          (64, 64, 128, 3072), (64, 128, 128, 1536), (64, 256, 256, 768), (64, 512, 512, 384)]:
    x = torch.randn(B, cin, L, device=dev); w = torch.randn(cout, cin, device=dev)
    w3 = w.unsqueeze(2).contiguous(); gy = torch.randn(B, cout, L, device=dev)
    fl = 2.0 * B * cin * cout * L
    if cin % 4 == 0:
        t1 = timeit(lambda: mfma_linear(gy, w, w_kmajor=True))
        t2 = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w3, None, [1], [0], [1], False, [0], 1, [True, False, False]))
        print("dgrad %-22s %10.3f %10.3f %8.2f | %10.1f %10.1f" % ("(%d,%d->%d,%d)" % (B, cin, cout, L), t1, t2, t2 / t1, fl / t1 / 1e9, fl / t2 / 1e9), flush=True)
    t1 = timeit(lambda: mfma_wgrad(x, gy, cout, cin, True))
    t2 = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w3, [cout], [1], [0], [1], False, [0], 1, [False, True, True]))
    print("wgrad %-22s %10.3f %10.3f %8.2f | %10.1f %10.1f" % ("(%d,%d->%d,%d)" % (B, cin, cout, L), t1, t2, t2 / t1, fl / t1 / 1e9, fl / t2 / 1e9), flush=True)
