"""Micro-benchmarks of the non-EMD ops (SURVEY 8d M2/M4/M5 shapes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvp_benchmark_amd import _lib as _lib0
if os.environ.get("MVP_LIB"):          # A/B: another build of the library (make variant ...)
    _lib0.LIB_PATH = os.path.abspath(os.environ["MVP_LIB"])
from mvp_benchmark_amd.metrics import cd
from mvp_benchmark_amd.mm3d_pn2 import (furthest_point_sample, knn, three_nn, three_interpolate,
                                        gather_points, grouping_operation, ball_query)
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else "all"

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best

def R(*shape):
    return torch.rand(*shape, generator=g).to(dev)

if which in ("all", "fps"):
    for (b, n, m) in [(64, 16384, 2048), (64, 2048, 512), (64, 2048, 2048), (64, 3072, 1536), (64, 1536, 768), (64, 768, 384), (32, 2048, 2048), (32, 3072, 1536), (64, 4096, 1024), (64, 8192, 1024)]:
        x = R(b, n, 3)
        ms = timeit(lambda: furthest_point_sample(x, m))
        print("fps (%d,%d)->%d: %.3f ms  %.3g sampled pts/s  %.3g updates/s  %.2f us/round" % (b, n, m, ms, b * m / ms * 1e3, b * (m - 1) * n / ms * 1e3, ms * 1e3 / (m - 1)), flush=True)
if which in ("all", "cd"):
    for (b, n, m) in [(64, 16384, 16384), (64, 2048, 16384), (32, 16384, 16384), (64, 2048, 2048), (64, 2048, 3072), (4, 2048, 2048)]:
        a, c = R(b, n, 3), R(b, m, 3)
        ms = timeit(lambda: cd()(a, c))
        print("cd (%d,%d)x(%d): %.3f ms  %.3g pair-evals/s (2 dirs)  %.1f TFLOP/s@16" % (b, n, m, ms, 2.0 * b * n * m / ms * 1e3, 16.0 * b * n * m / ms * 1e3 / 1e12), flush=True)
if which in ("all", "pn2"):
    for (b, n, k) in [(64, 2048, 16), (64, 16384, 16), (64, 2048, 10), (64, 2048, 20)]:
        x = R(b, n, 3)
        ms = timeit(lambda: knn(k, x, x, False), 3)
        print("knn k=%d (%d,%d): %.3f ms  %.3g evals/s" % (k, b, n, ms, 1.0 * b * n * n / ms * 1e3), flush=True)
    for (b, n, m, c) in [(64, 768, 384, 512), (64, 1536, 768, 256), (64, 3072, 1536, 128)]:
        tgt, src, f = R(b, n, 3), R(b, m, 3), R(b, c, m)
        ms = timeit(lambda: three_nn(tgt, src))
        dist, idx = three_nn(tgt, src)
        w = torch.rand(b, n, 3, device=dev)
        ms2 = timeit(lambda: three_interpolate(f, idx, w))
        print("three_nn (%d,%d<-%d): %.3f ms; three_interpolate C=%d: %.3f ms  %.1f GB/s (8 B/out elem)" % (b, n, m, ms, c, ms2, 8.0 * b * c * n / ms2 / 1e6), flush=True)
    for (b, c, n, m) in [(64, 3, 3072, 1536), (64, 64, 3072, 15360), (64, 128, 1536, 7680), (64, 256, 768, 3840)]:
        f = R(b, c, n); idx = torch.randint(0, n, (b, m), generator=g, dtype=torch.int32).to(dev)
        ms = timeit(lambda: gather_points(f, idx))
        print("gather (%d,%d,%d)->%d: %.3f ms  %.1f GB/s (8 B/out elem)" % (b, c, n, m, ms, 8.0 * b * c * m / ms / 1e6), flush=True)
    for (n, m, r, s) in [(1024, 51, 0.0632, 4), (2048, 102, 0.1095, 24)]:
        x = R(64, n, 3); ctr = x[:, :m].contiguous()
        ms = timeit(lambda: ball_query(0.0, r, s, x, ctr))
        print("ball_query (64,%d) M=%d S=%d: %.3f ms" % (n, m, s, ms), flush=True)
if which in ("all", "grad"):
    for (b, c, n, m) in [(64, 64, 3072, 49152), (64, 128, 1536, 24576), (64, 256, 768, 12288), (64, 512, 384, 6144)]:
        f = R(b, c, n).requires_grad_()
        idx = torch.randint(0, n, (b, m), generator=g).int().to(dev)
        out = gather_points(f, idx)
        go = torch.rand_like(out)
        ms = timeit(lambda: torch.autograd.grad(out, f, go, retain_graph=True))
        print("gather grad (%d,%d,%d)<-%d: %.3f ms  %.1f GB/s of grad_out read" % (b, c, n, m, ms, 4.0 * b * c * m / ms / 1e6), flush=True)
    for (b, c, m, n) in [(32, 512, 1024, 3072), (32, 768, 256, 1024), (64, 128, 1536, 3072)]:
        f = R(b, c, m).requires_grad_()
        idx = torch.randint(0, m, (b, n, 3), generator=g).int().to(dev)
        w = R(b, n, 3)
        out = three_interpolate(f, idx, w)
        go = torch.rand_like(out)
        ms = timeit(lambda: torch.autograd.grad(out, f, go, retain_graph=True))
        print("three_interpolate grad (%d,%d,%d)<-%d: %.3f ms  %.1f GB/s of grad_out read" % (b, c, m, n, ms, 4.0 * b * c * n / ms / 1e6), flush=True)
if which in ("all", "grad", "cdgrad"):
    from mvp_benchmark_amd import _lib
    for (b, n, m) in [(64, 2048, 2048), (64, 2048, 3072), (64, 8192, 8192)]:
        x1, x2 = R(b, n, 3), R(b, m, 3)
        d1 = torch.empty(b, n, device=dev); d2 = torch.empty(b, m, device=dev)
        i1 = torch.empty(b, n, dtype=torch.int32, device=dev); i2 = torch.empty(b, m, dtype=torch.int32, device=dev)
        _lib.call("mvp_chamfer_forward", dev, b, n, m, x1, x2, d1, d2, i1, i2)
        g1, g2 = R(b, n), R(b, m)
        gx1, gx2 = torch.zeros(b, n, 3, device=dev), torch.zeros(b, m, 3, device=dev)
        ms = timeit(lambda: _lib.call("mvp_chamfer_backward", dev, b, n, m, x1, x2, gx1, gx2, g1, g2, i1, i2))
        print("cd backward (%d,%d)x(%d): %.3f ms" % (b, n, m, ms), flush=True)
if which in ("all", "topk"):
    from mvp_benchmark_amd.mm3d_pn2.functional import gram_topk
    for (b, c, n, k) in [(32, 24, 3072, 16), (32, 48, 1024, 16), (32, 48, 256, 16)]:
        x = R(b, c, n)
        dot = torch.matmul(x.transpose(2, 1), x).contiguous()
        sq = (x * x).sum(1).contiguous()
        ms = timeit(lambda: gram_topk(dot, sq, k))
        print("gram top-k (%d,%d,%d) k=%d: %.3f ms  %.1f GB/s of the Gram matrix" % (b, n, n, k, ms, 4.0 * b * n * n / ms / 1e6), flush=True)
if which in ("all", "harness"):
    # The reference's own two timing harnesses, as it runs them:
    #  * utils/metrics/CD/unit_test.py:38-62 `timings()`: 100 iterations of chamfer forward + backward at (32, 2000, 3) /
    #    (32, 1000, 3), loss = d1.sum() + d2.sum() (it prints seconds per 100 iterations; nothing is recorded in the repo)
    #  * utils/metrics/EMD/emd_module.py:90-104 `test_emd()`: ONE call at (20, 8192, 3), eps 0.05, 3000 iterations
    import time
    from mvp_benchmark_amd.metrics import emd
    p1, p2 = R(32, 2000, 3).requires_grad_(), R(32, 1000, 3)
    cham = cd()
    def cd_iter():
        d1, d2, _, _ = cham(p1, p2)
        (d1.sum() + d2.sum()).backward()
    for _ in range(3):
        cd_iter()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100):
        cd_iter()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("reference harness CD (unit_test.py:38-62): 100 x forward+backward at (32,2000,3)/(32,1000,3): %.4f s total, %.3f ms / iteration" % (t1 - t0, (t1 - t0) * 10), flush=True)
    x1, x2 = R(20, 8192, 3), R(20, 8192, 3)
    e = emd()
    ms = timeit(lambda: e(x1, x2, 0.05, 3000), 3)
    dist, ass = e(x1, x2, 0.05, 3000)
    print("reference harness EMD (emd_module.py:90-104): (20,8192,3) eps 0.05, 3000 iterations: %.3f ms; mean sqrt(dist) %.5f; distinct targets %d of %d"
          % (ms, dist.sqrt().mean().item(), int(torch.unique(ass[0]).numel()), ass.size(1)), flush=True)
