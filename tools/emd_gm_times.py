"""Phase clock of the gathered-bid rounds (library built with -DMVP_EMD_GMTIME:
   make -C mvp_benchmark_amd/csrc variant NAME=gmt DEFS=-DMVP_EMD_GMTIME FILES=emd_lean.hip).
   python tools/emd_gm_times.py lib.so [B N]   ->  per cloud (of the launch that finished it): cycles per round and phase,
   as wave 0 of member 0 sees them, the cost of a stamp (~half of `cal`) not subtracted."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mvp_benchmark_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
b = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
x1, x2 = torch.rand(b, n, 3, generator=g).to(dev), torch.rand(b, n, 3, generator=g).to(dev)
shape = os.environ.get("MVP_BENCH_SHAPE")   # e.g. "chair:0.03" (gt + noise), "chair:indep": tools/emd_surfaces.py's clouds
if shape:
    from mvp_benchmark_amd.synthetic import prediction_pair
    name, _, mode = shape.partition(":")
    pred, gt = prediction_pair(name, mode or "indep", g, b, n)
    x1, x2 = pred.to(dev), gt.to(dev)
nbytes = _lib.emd_scratch_bytes(b, n)
scratch = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
dist = torch.zeros(b, n, device=dev)
ass = torch.zeros(b, n, dtype=torch.int32, device=dev)
for _ in range(2):
    _lib.call("mvp_emd_forward", dev, b, n, x1, x2, dist, ass, 0.004, 3000, scratch, nbytes)
torch.cuda.synchronize()
per_cloud = n * (68 + 8 * 8) + 8 * 2048 * 8 + (1728 + 4) * 4
names = ["top", "bid", "words", "bids", "decode", "barrier", "contest", "apply", "drain", "closing", "counts", "cal"]
idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 14]
rows = []
for c in range(b):
    off = c * per_cloud + n * (68 + 64) + (8 * 2048 - 128) * 8
    v = scratch[off: off + 17 * 8].view(torch.int64).cpu().numpy()
    if v[15] > 0:
        rows.append((c, int(v[16]), int(v[15])) + tuple(v[i] / v[15] for i in idx))
rows.sort(key=lambda r: -sum(r[3:14]))
print("cloud  W rounds | " + " ".join("%7s" % s for s in names) + " |   total (cycles per gathered-bid round)")
for r in rows[:6] + rows[-3:]:
    print("%5d %2d %6d | " % r[:3] + " ".join("%7.0f" % x for x in r[3:]) + " | %7.0f" % sum(r[3:14]))
m = np.mean([r[3:] for r in rows], axis=0)
print("mean            | " + " ".join("%7.0f" % x for x in m) + " | %7.0f" % m[:11].sum())
