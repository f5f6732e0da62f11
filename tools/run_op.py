"""Run one operator at a benchmark shape a few times (for rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvp_benchmark_amd.metrics import cd, emd
from mvp_benchmark_amd.mm3d_pn2 import furthest_point_sample

op = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
n = int(sys.argv[4]) if len(sys.argv) > 4 else 16384
g = torch.Generator().manual_seed(1000)
a = torch.rand(B, n, 3, generator=g).cuda()
b = torch.rand(B, n, 3, generator=g).cuda()
for _ in range(reps):
    if op == "emd":
        emd()(a, b, 0.004, 3000)
    elif op == "cd":
        cd()(b, a)
    elif op == "fps":
        furthest_point_sample(a, 2048)
torch.cuda.synchronize()
print("done", op)
