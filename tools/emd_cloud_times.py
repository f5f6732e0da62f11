"""Per-cloud, per-launch times of the lean kernel's launches (library built with -DMVP_EMD_CLOUDTIME:
   make -C mvp_benchmark_amd/csrc variant NAME=ctime DEFS=-DMVP_EMD_CLOUDTIME FILES=emd_lean.hip).
   python tools/emd_cloud_times.py lib.so [B N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mvp_benchmark_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
b, n = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (64, 16384)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x1 = torch.rand(b, n, 3, generator=g).to(dev); x2 = torch.rand(b, n, 3, generator=g).to(dev)
nbytes = _lib.emd_scratch_bytes(b, n)
scratch = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
dist = torch.zeros(b, n, device=dev); ass = torch.zeros(b, n, dtype=torch.int32, device=dev)
for _ in range(2):
    scratch.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _lib.call("mvp_emd_forward", dev, b, n, x1, x2, dist, ass, 0.004, 3000, scratch, nbytes); e1.record()
    torch.cuda.synchronize()
print("call: %.2f ms" % e0.elapsed_time(e1))
per = n * 132 + 131072 + (1728 + 4) * 4
raw = scratch.cpu().numpy()
rows = {}
for c in range(b):
    dbg = raw[c * per + n * 132 + 131072 - 512: c * per + n * 132 + 131072].view(np.uint64)
    for k in range(64):
        v = int(dbg[k])
        if v:
            rows.setdefault(k, []).append((c, v >> 48, (v >> 32) & 0xFFFF, (v & 0xFFFFFFFF) * 0.01))
for k in sorted(rows):
    a = np.array(rows[k])
    name = "last launch" if k == 63 else "first kernel (to the hand-over)" if k == 62 else "launch ending at round ~%d" % (k * 64)
    print("%s: %d clouds, time mean %.0f max %.0f us" % (name, len(a), a[:, 3].mean(), a[:, 3].max()))
    for w in sorted(set(a[:, 1].astype(int))):
        z = a[a[:, 1] == w]
        print("    W=%d: %2d clouds  time mean %6.0f min %6.0f max %6.0f us   unassigned at the end mean %4.0f" % (w, len(z), z[:, 3].mean(), z[:, 3].min(), z[:, 3].max(), z[:, 2].mean()))
