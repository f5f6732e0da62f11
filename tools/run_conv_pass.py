"""Run the three passes of one 1x1-convolution shape a few times (for rocprofv3 --pmc / --kernel-trace).
python tools/run_conv_pass.py B Cin Cout L [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvp_benchmark_amd import _lib
if os.environ.get('MVP_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['MVP_LIB'])
from mvp_benchmark_amd.pointwise import mfma_linear, mfma_wgrad
B, cin, cout, L = (int(a) for a in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = "cuda:0"
x = torch.randn(B, cin, L, device=dev); w = torch.randn(cout, cin, device=dev); b = torch.randn(cout, device=dev)
gy = torch.randn(B, cout, L, device=dev)
y = mfma_linear(x, w, b, relu=True)
for _ in range(reps):
    mfma_linear(x, w, b, relu=True)
    mfma_linear(gy, w, w_kmajor=True, xmask=y)
    mfma_wgrad(x, gy, cout, cin, True, gymask=y)
    mfma_wgrad(x, gy, cout, cin, True)
torch.cuda.synchronize()
print("done")
