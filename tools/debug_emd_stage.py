import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mvp_benchmark_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
b, n = 1, 1024
rng = np.random.default_rng(0)
dev = torch.device("cuda:0")
t1 = torch.tensor(rng.random((b, n, 3), dtype=np.float32), device=dev)
t2 = torch.tensor(rng.random((b, n, 3), dtype=np.float32), device=dev)
nbytes = _lib.emd_scratch_bytes(b, n)
scratch = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
dist = torch.zeros(b, n, device=dev); ass = torch.zeros(b, n, dtype=torch.int32, device=dev)
_lib.call("mvp_emd_forward", dev, b, n, t1, t2, dist, ass, 0.005, 1, scratch, nbytes)
torch.cuda.synchronize()
print(sys.argv[1], "returned; stats", scratch[nbytes - b * 16:].view(torch.int64).tolist(), flush=True)
