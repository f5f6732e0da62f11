"""GPU report (not collected by pytest): the REFERENCE's own kernels (oracle/_ref, built by oracle/build_ref_gpu.sh) next to
the CPU oracle and this repo's HIP path -- how many results differ, and (with --time) how long each takes on the box.

    python tests/report_reference_kernels.py [--time] [--variant _nofma]

Lives under tests/ because it executes oracle/ code (test infrastructure only).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as orc  # noqa: E402
import ref_kernels as ref  # noqa: E402
from mvp_benchmark_amd import metrics, mm3d_pn2 as pn2  # noqa: E402


def rnd(seed, *shape):
    return np.random.default_rng(seed).random(shape, dtype=np.float32)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def diff(name, a, b, what="vs oracle"):
    a = a.cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    n = int((a != b).sum())
    extra = ""
    if n and a.dtype.kind == "f":
        extra = f"  max |d| {np.abs(a.astype(np.float64) - b).max():.3g}  max rel {np.max(np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(b), 1e-30)):.3g}"
    print(f"  {name:34s} {what}: {n} of {a.size} differ{extra}", flush=True)
    return n


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="")
    ap.add_argument("--time", action="store_true")
    args = ap.parse_args()
    v = args.variant
    orc.build()
    orc.set_contraction(v != "_nofma")   # the oracle in the arithmetic the variant was compiled with
    print(f"reference kernels variant '{v or 'default contraction'}' on {torch.cuda.get_device_name(0)}")

    print("FPS")
    for (b, n, m) in [(4, 2048, 512), (3, 1000, 300), (2, 16384, 2048), (5, 100, 37), (2, 8193, 1024), (2, 513, 200)]:
        x = rnd(n + m, b, n, 3)
        r = ref.fps(dev(x), m, v)
        diff(f"fps ({b},{n})->{m} ref", r, orc.furthest_point_sample(x, m))
        diff(f"fps ({b},{n})->{m} ours", pn2.furthest_point_sample(dev(x), m), r, "vs reference kernel")
    x = rnd(5, 3, 700, 3)
    d = ((x[:, :, None, :] - x[:, None, :, :]) ** 2).sum(-1).astype(np.float32)
    r = ref.fps_with_dist(dev(d), 128, v)
    diff("fps_with_dist (3,700)->128 ref", r, orc.furthest_point_sample_with_dist(d, 128))
    diff("fps_with_dist ours", pn2.furthest_point_sample_with_dist(dev(d), 128), r, "vs reference kernel")
    # lattice: many exact ties
    g = np.stack(np.meshgrid(*[np.arange(8, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(1, -1, 3) / 8
    r = ref.fps(dev(g), 200, v)
    diff("fps lattice 512->200 ref", r, orc.furthest_point_sample(g, 200))
    diff("fps lattice ours", pn2.furthest_point_sample(dev(g), 200), r, "vs reference kernel")

    print("queries")
    xyz, ctr = rnd(1, 4, 2048, 3), rnd(2, 4, 512, 3)
    r = ref.ball_query(0.0, 0.2, 32, dev(xyz), dev(ctr), v)
    diff("ball_query r=.2 s=32 ref", r, orc.ball_query(0.0, 0.2, 32, xyz, ctr))
    diff("ball_query ours", pn2.ball_query(0.0, 0.2, 32, dev(xyz), dev(ctr)), r, "vs reference kernel")
    r = ref.ball_query(0.05, 0.3, 16, dev(xyz), dev(ctr), v)
    diff("ball_query .05-.3 s=16 ref", r, orc.ball_query(0.05, 0.3, 16, xyz, ctr))
    for k in (1, 8, 16, 20):
        ri, rd = ref.knn(k, dev(xyz), dev(ctr), v)
        oi, od = orc.knn(k, xyz, ctr, return_dist=True)
        diff(f"knn k={k} idx ref", ri.transpose(2, 1), oi)
        diff(f"knn k={k} dist2 ref", rd, od)
        diff(f"knn k={k} ours", pn2.knn(k, dev(xyz), dev(ctr)), ri.transpose(2, 1).contiguous(), "vs reference kernel")
    rd, ri = ref.three_nn(dev(ctr), dev(xyz), v)
    od, oi = orc.three_nn(ctr, xyz)
    diff("three_nn idx ref", ri, oi)
    diff("three_nn sqrt(dist2) ref", np.sqrt(rd.cpu().numpy()), od)
    md, mi = pn2.three_nn(dev(ctr), dev(xyz))
    diff("three_nn idx ours", mi, ri, "vs reference kernel")

    print("gather / group / interpolate")
    feat = rnd(3, 4, 24, 2048)
    idx = np.random.default_rng(4).integers(0, 2048, (4, 300)).astype(np.int32)
    r = ref.gather_points(dev(feat), dev(idx), v)
    diff("gather ref", r, orc.gather_points(feat, idx))
    diff("gather ours", pn2.gather_points(dev(feat), dev(idx)), r, "vs reference kernel")
    go = rnd(5, 4, 24, 300)
    r = ref.gather_points_grad(dev(go), dev(idx), 2048, v)
    o = orc.gather_points_grad(go, idx, 2048)
    diff("gather_grad ref (float atomics)", r, o)
    gidx = np.random.default_rng(6).integers(0, 2048, (4, 128, 16)).astype(np.int32)
    r = ref.grouping_operation(dev(feat), dev(gidx), v)
    diff("group ref", r, orc.grouping_operation(feat, gidx))
    diff("group ours", pn2.grouping_operation(dev(feat), dev(gidx)), r, "vs reference kernel")
    gg = rnd(7, 4, 24, 128, 16)
    r = ref.grouping_operation_grad(dev(gg), dev(gidx), 2048, v)
    diff("group_grad ref (float atomics)", r, orc.grouping_operation_grad(gg, gidx, 2048))
    w = rnd(8, 4, 512, 3)
    w /= w.sum(-1, keepdims=True)
    f512 = rnd(9, 4, 24, 2048)
    r = ref.three_interpolate(dev(f512), ri, dev(w), v)
    diff("three_interpolate ref", r, orc.three_interpolate(f512, ri.cpu().numpy(), w))
    diff("three_interpolate ours", pn2.three_interpolate(dev(f512), ri, dev(w)), r, "vs reference kernel")
    gi = rnd(10, 4, 24, 512)
    r = ref.three_interpolate_grad(dev(gi), ri, dev(w), 2048, v)
    diff("three_interpolate_grad ref (atomics)", r, orc.three_interpolate_grad(gi, ri.cpu().numpy(), w, 2048))

    print("chamfer")
    for (b, n, m) in [(4, 100, 200), (2, 2048, 2048), (2, 2048, 16384), (3, 777, 1300)]:
        a, c = rnd(n, b, n, 3), rnd(m + 1, b, m, 3)
        d1, d2, i1, i2 = ref.chamfer_forward(dev(a), dev(c), v)
        o = orc.chamfer_forward(a, c)
        for nm, rr, oo in zip(("dist1", "dist2", "idx1", "idx2"), (d1, d2, i1, i2), o):
            diff(f"chamfer ({b},{n},{m}) {nm} ref", rr, oo)
        m1, m2, j1, j2 = metrics.cd()(dev(a), dev(c))
        for nm, rr, oo in zip(("dist1", "dist2", "idx1", "idx2"), (m1, m2, j1, j2), (d1, d2, i1, i2)):
            diff(f"chamfer ({b},{n},{m}) {nm} ours", rr, oo, "vs reference kernel")
    g1, g2 = rnd(11, 3, 777), rnd(12, 3, 1300)
    gx1, gx2 = ref.chamfer_backward(dev(a), dev(c), dev(g1), dev(g2), i1, i2, v)
    o1, o2 = orc.chamfer_backward(a, c, g1, g2, i1.cpu().numpy(), i2.cpu().numpy())
    diff("chamfer_backward gradxyz1 ref (atomics)", gx1, o1)
    diff("chamfer_backward gradxyz2 ref (atomics)", gx2, o2)

    print("EMD")
    for (b, n, eps, it) in [(2, 1024, 0.005, 50), (2, 1024, 0.002, 10000), (4, 2048, 0.004, 3000), (2, 8192, 0.004, 3000),
                            (2, 2048, 0.05, 100), (2, 4096, 0.004, 3000)]:
        a, c = rnd(n + it, b, n, 3), rnd(n + it + 1, b, n, 3)
        runs = [ref.emd_forward(dev(a), dev(c), eps, it, v) for _ in range(3)]
        od, oa = orc.emd_forward(a, c, eps, it)
        same_runs = all(torch.equal(runs[0][1], r[1]) for r in runs[1:])
        lo = orc.emd_forward_ex(a, c, eps, it, getmax_lowest=True)
        policy_free = bool(np.array_equal(np.asarray(lo[1]), oa))
        print(f"  emd ({b},{n}) eps {eps} iters {it}: reference kernel run-to-run identical: {same_runs}; "
              f"oracle independent of the GetMax winner policy: {policy_free}")
        if not same_runs:
            diff("    assignment run 0 vs run 1", runs[0][1], runs[1][1], "(reference kernel vs itself)")
        diff("    assignment ref", runs[0][1], oa)
        diff("    dist ref", runs[0][0], od)
        md, ma = metrics.emd()(dev(a), dev(c), eps, it)
        diff("    assignment ours", ma, runs[0][1], "vs reference kernel")
        diff("    dist ours", md, runs[0][0], "vs reference kernel")
        print(f"    sqrt(mean dist): reference kernel {float(runs[0][0].sqrt().mean()):.8f}  oracle {float(np.sqrt(od).mean()):.8f}  ours {float(md.sqrt().mean()):.8f}")

    if args.time:
        print("timings, ms (reference kernel as compiled for gfx950 | this repo)")
        a, c = dev(rnd(1, 64, 16384, 3)), dev(rnd(2, 64, 16384, 3))
        cdm = metrics.cd()
        print(f"  chamfer fwd (64,16384,16384):  {timed(lambda: ref.chamfer_forward(a, c, v)):9.2f} | {timed(lambda: cdm(a, c)):8.2f}")
        x = dev(rnd(3, 64, 16384, 3))
        print(f"  fps (64,16384)->2048:          {timed(lambda: ref.fps(x, 2048, v), 3):9.2f} | {timed(lambda: pn2.furthest_point_sample(x, 2048), 3):8.2f}")
        x2, c2 = dev(rnd(4, 64, 2048, 3)), dev(rnd(5, 64, 512, 3))
        print(f"  fps (64,2048)->512:            {timed(lambda: ref.fps(x2, 512, v)):9.2f} | {timed(lambda: pn2.furthest_point_sample(x2, 512)):8.2f}")
        print(f"  knn k=16 (64,2048) q 2048:     {timed(lambda: ref.knn(16, x2, x2, v)):9.2f} | {timed(lambda: pn2.knn(16, x2, x2)):8.2f}")
        print(f"  ball_query s=32 (64,2048,512): {timed(lambda: ref.ball_query(0., .2, 32, x2, c2, v)):9.2f} | {timed(lambda: pn2.ball_query(0., .2, 32, x2, c2)):8.2f}")
        print(f"  three_nn (64,2048<-512):       {timed(lambda: ref.three_nn(x2, c2, v)):9.2f} | {timed(lambda: pn2.three_nn(x2, c2)):8.2f}")
        f1 = dev(rnd(8, 64, 128, 2048))
        gi = dev(np.random.default_rng(9).integers(0, 2048, (64, 512, 32)).astype(np.int32))
        print(f"  group (64,128,2048) idx (512,32): {timed(lambda: ref.grouping_operation(f1, gi, v)):7.2f} | {timed(lambda: pn2.grouping_operation(f1, gi)):8.2f}")
        ti = dev(np.random.default_rng(10).integers(0, 512, (64, 2048, 3)).astype(np.int32))
        tw = dev(rnd(11, 64, 2048, 3))
        f2 = dev(rnd(12, 64, 256, 512))
        print(f"  three_interpolate (64,256,512->2048): {timed(lambda: ref.three_interpolate(f2, ti, tw, v)):5.2f} | {timed(lambda: pn2.three_interpolate(f2, ti, tw)):8.2f}")
        # the reference's own harness shape for CD (unit_test.py / SURVEY 6): forward + backward at (32,2000,3)/(32,1000,3)
        ha, hc = dev(rnd(13, 32, 2000, 3)), dev(rnd(14, 32, 1000, 3))
        g1, g2 = dev(rnd(15, 32, 2000)), dev(rnd(16, 32, 1000))

        def ref_cd_fb():
            d1, d2, i1, i2 = ref.chamfer_forward(ha, hc, v)
            ref.chamfer_backward(ha, hc, g1, g2, i1, i2, v)

        def our_cd_fb():
            pa, pc = ha.clone().requires_grad_(True), hc.clone().requires_grad_(True)
            d1, d2, _, _ = cdm(pa, pc)
            ((d1 * g1).sum() + (d2 * g2).sum()).backward()

        print(f"  chamfer fwd+bwd (32,2000)/(32,1000): {timed(ref_cd_fb, 20):6.3f} | {timed(our_cd_fb, 20):8.3f}")
        em = metrics.emd()
        step_ref = step_our = None
        for (b, n, eps, it, reps) in [(20, 8192, 0.05, 3000, 1), (64, 2048, 0.005, 50, 3), (64, 1024, 0.004, 3000, 1),
                                      (64, 2048, 0.004, 3000, 1), (64, 4096, 0.004, 3000, 1), (64, 8192, 0.004, 3000, 1),
                                      (64, 16384, 0.004, 3000, 1)]:
            p, q = dev(rnd(6, b, n, 3)), dev(rnd(7, b, n, 3))
            tr = timed(lambda: ref.emd_forward(p, q, eps, it, v), reps)
            to = timed(lambda: em(p, q, eps, it), reps)
            print(f"  emd ({b},{n}) eps {eps} iters {it}: {tr:9.1f} | {to:8.2f}   ({tr / to:.0f}x)", flush=True)
            step_ref, step_our = tr, to
        tcr, tco = timed(lambda: ref.chamfer_forward(a, c, v)), timed(lambda: cdm(a, c))
        print(f"  headline step (CD + EMD at (64,16384,3), eps 0.004, 3000 rounds): {tcr + step_ref:9.1f} | {tco + step_our:8.2f}   ({(tcr + step_ref) / (tco + step_our):.0f}x)")
        from mvp_benchmark_amd import synthetic
        for shape, modev in [("chair", 0.03), ("chair", "indep"), ("sphere", 0.03)]:
            g = torch.Generator().manual_seed(5)
            pr, gt = synthetic.prediction_pair(shape, modev, g, 64, 16384)
            pr, gt = pr.cuda(), gt.cuda()
            tr = timed(lambda: ref.emd_forward(pr, gt, 0.004, 3000, v), 1)
            to = timed(lambda: em(pr, gt, 0.004, 3000), 1)
            print(f"  emd surface {shape} / {modev} (64,16384): {tr:9.1f} | {to:8.2f}   ({tr / to:.0f}x)", flush=True)


if __name__ == "__main__":
    main()
