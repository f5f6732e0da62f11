import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvp_benchmark_amd import _lib
libs = {"A": _lib.LIB_PATH, "B": os.path.abspath(sys.argv[1])}
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
handles = {}
for k, p in libs.items():
    h = ctypes.CDLL(p)
    fn = h.mvp_furthest_point_sampling
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 4
    handles[k] = fn
for (b, n, m) in [(64, 16384, 2048), (64, 2048, 2048), (64, 768, 384)]:
    x = torch.rand(b, n, 3, generator=g).to(dev)
    temp = torch.empty(b, n, device=dev); idx = torch.zeros(b, m, dtype=torch.int32, device=dev)
    res = {k: [] for k in handles}
    for rep in range(6):
        for k, fn in handles.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(b, n, m, x.data_ptr(), temp.data_ptr(), idx.data_ptr(), torch.cuda.current_stream().cuda_stream)
            e1.record(); torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1))
    print((b, n, m), {k: "min %.3f med %.3f" % (min(v), sorted(v)[len(v) // 2]) for k, v in res.items()}, flush=True)
