"""One EMD shape: python tools/bench_emd_one.py B N EPS ITERS [lib.so]  (MVP_EMD_CLUSTER selects the cluster width)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvp_benchmark_amd import _lib
b, n, eps, iters = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
if len(sys.argv) > 5:
    _lib.LIB_PATH = os.path.abspath(sys.argv[5])
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x1 = torch.rand(b, n, 3, generator=g).to(dev); x2 = torch.rand(b, n, 3, generator=g).to(dev)
shape = os.environ.get("MVP_BENCH_SHAPE")   # e.g. "sphere", "chair:0.03" (gt + noise), "torus:indep": tools/emd_surfaces.py's clouds
if shape:
    from mvp_benchmark_amd.synthetic import prediction_pair
    name, _, mode = shape.partition(":")
    pred, gt = prediction_pair(name, mode or "indep", g, b, n)
    x1, x2 = pred.to(dev), gt.to(dev)
nbytes = _lib.emd_scratch_bytes(b, n)
scratch = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
dist = torch.zeros(b, n, device=dev); ass = torch.zeros(b, n, dtype=torch.int32, device=dev)
best = 1e9
for rep in range(int(os.environ.get("MVP_BENCH_REPS", 3 if len(sys.argv) <= 5 else 1))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.call("mvp_emd_forward", dev, b, n, x1, x2, dist, ass, eps, iters, scratch, nbytes)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
st = scratch[nbytes - b * 16:].view(torch.int64).view(b, 2).cpu()
print("W=%s b=%d n=%d eps=%g iters=%d: %.2f ms rounds %d bids/cloud %.0f" % (
    os.environ.get("MVP_EMD_CLUSTER", "auto"), b, n, eps, iters, best, int(st[:, 0].max()), st[:, 1].double().mean().item()), flush=True)
