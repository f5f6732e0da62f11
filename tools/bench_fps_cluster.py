"""FPS with W workgroups per cloud (mvp_furthest_point_sampling_cluster) against the one-workgroup kernels
(register-resident / Morton-sorted), at the sizes where a lane owns many points.  python tools/bench_fps_cluster.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvp_benchmark_amd import _lib
from mvp_benchmark_amd.mm3d_pn2 import furthest_point_sample

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for (b, n, m) in [(64, 16384, 2048), (32, 16384, 2048), (64, 8192, 2048), (64, 4096, 1024), (64, 2048, 512), (128, 8192, 1024)]:
    x = torch.rand(b, n, 3, generator=g).to(dev)
    base = timeit(lambda: furthest_point_sample(x, m))
    ref = furthest_point_sample(x, m)
    line = "fps (%d, %d -> %d): one workgroup per cloud %.3f ms (%.2f us/round)" % (b, n, m, base, base * 1e3 / (m - 1))
    for w in (2, 4):
        if n % (w * 1024):
            continue
        nbytes = _lib.fps_cluster_scratch_bytes(b)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        temp = torch.empty(b, n, device=dev)
        idx = torch.zeros(b, m, dtype=torch.int32, device=dev)

        def run():
            temp.fill_(1e10)
            _lib.call("mvp_furthest_point_sampling_cluster", dev, b, n, m, w, x, temp, idx, scratch, nbytes)
        try:
            ms = timeit(run)
            ok = torch.equal(idx, ref)
            line += "; %d workgroups per cloud %.3f ms (%.2f us/round%s)" % (w, ms, ms * 1e3 / (m - 1), "" if ok else ", MISMATCH")
        except Exception as e:  # noqa: BLE001
            line += "; W=%d: %s" % (w, str(e)[:60])
    print(line, flush=True)
