"""Within-run A/B of EMD library variants: python tools/ab_emd.py lib1.so lib2.so ..."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvp_benchmark_amd import _lib
libs = {"base": _lib.LIB_PATH}
for p in sys.argv[1:]:
    libs[os.path.basename(p)] = os.path.abspath(p)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
fns = {}
for k, p in libs.items():
    h = ctypes.CDLL(p)
    fn = h.mvp_emd_forward
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
    fns[k] = fn
for (b, n, eps, iters) in [(64, 1024, 0.004, 3000), (64, 4096, 0.004, 3000), (64, 16384, 0.004, 3000), (64, 16384, 0.005, 50)]:
    x1 = torch.rand(b, n, 3, generator=g).to(dev); x2 = torch.rand(b, n, 3, generator=g).to(dev)
    nbytes = _lib.emd_scratch_bytes(b, n)
    scratch = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    dist = torch.zeros(b, n, device=dev); ass = torch.zeros(b, n, dtype=torch.int32, device=dev)
    res = {k: [] for k in fns}
    ref = None
    for rep in range(3):
        for k, fn in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(b, n, x1.data_ptr(), x2.data_ptr(), dist.data_ptr(), ass.data_ptr(), eps, iters, scratch.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
            e1.record(); torch.cuda.synchronize()
            assert rc == 0
            res[k].append(e0.elapsed_time(e1))
            if ref is None:
                ref = ass.clone()
            else:
                assert torch.equal(ref, ass), "variants disagree"
    print((b, n, eps, iters), {k: "%.2f" % min(v) for k, v in res.items()}, flush=True)
