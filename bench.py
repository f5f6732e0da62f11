#!/usr/bin/env python
"""Headline benchmark of the MI355X-native MVP op layer.

A "step" = one pass of the completion-eval hot path over one synthetic batch:
    calc_cd(pred, gt, calc_f1=True)  -> cd() + fscore + the cd_p/cd_t reductions
    calc_emd(pred, gt, eps=0.004, iterations=3000)
on pred, gt = (64, 16384, 3) uniform [0,1) clouds per GPU (BASELINE.json
metric: "point-pairs/sec CD+EMD @2048->16384 pts, batch 64").  Inputs are
resident in HBM before the timed region.  value = B*N*M*n_gpus / time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload eval|vrcnet_train|pcn_eval]

N > 1: one rank per GPU over RCCL.  Started by the driver under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the
script reads RANK / LOCAL_RANK / WORLD_SIZE and refuses to run unless
WORLD_SIZE == N; started bare (`python bench.py --gpus N`) it re-executes
itself under torch.distributed.run with N local ranks -- and fails loudly if the
box has fewer than N GPUs, so a "--gpus 8" line can never be a 1-GPU number.
The batch dimension is sharded (every rank evaluates its own 64 clouds: weak
scaling); the only collective of the eval workload is the 5-float metric
all-reduce of the eval loop.

--workload vrcnet_train is BASELINE cfg 3 (completion/train.py:122-142 with
cfgs/vrcnet.yaml): DDP training steps of VRCNet, 32 clouds per rank (global
batch 32*N), CD loss, Adam; its line reports samples/s, steps/s, the gradient
all-reduce volume per step and the RCCL bus bandwidth of an all-reduce of that
volume measured beside the timed region.

--workload pcn_eval is BASELINE cfg 2 / SURVEY M6 (completion/test.py:23-64, train.py's val(), models/pcn.py:75-112
with cfgs/pcn_eval16k.yaml): the eval step of PCN at 2048 -> 16384 points, 32 clouds per rank -- network forward,
calc_cd with F1 on the network's output, calc_emd at the eval setting -- with per-part milliseconds.

Rank 0 prints ONE JSON line (contract in the task statement) including
  roofline     -- for the dominant kernel (the persistent EMD auction kernel)
  cpu_baseline -- the CPU oracle timed on a bounded sample of the same workload
  cpu_baseline_reference_path -- the reference's own CPU path for CD
  gpu_reference_baseline -- the reference's own CD / EMD kernels (oracle/_ref, compiled for gfx950 as written) on the timed batch
                  (distChamfer, restated in metrics/CD/chamfer_python.py) timed
                  with all torch threads
  extra        -- per-op times, FPS throughput, CD figures.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# non-packed FP32 VALU issue roof: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz (MI355X_MICROARCH.md)
VALU_LANE_OPS_PER_S = 256 * 4 * 16 * 2.4e9


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=("eval", "vrcnet_train", "pcn_eval"), default="eval")
    ap.add_argument("--batch", type=int, default=None, help="clouds per GPU (eval: 64, vrcnet_train / pcn_eval: 32)")
    ap.add_argument("--points", type=int, default=16384)
    ap.add_argument("--eps", type=float, default=0.004)
    ap.add_argument("--iters", type=int, default=3000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true",
                    help="skip the untimed side measurements (tools/profile_round.sh: keeps the kernel trace to the headline shape)")
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="clouds in the CPU-baseline sample (0 = min(cores, 64))")
    args = ap.parse_args()
    if args.steps is None:
        # (~7 s of timed region for the eval step: long enough for a 5-s utilisation sampler to see it)
        args.steps = 150 if args.workload == "eval" else 50 if args.workload == "pcn_eval" else 10
    if args.warmup is None:
        args.warmup = 3
    if args.batch is None:
        args.batch = 64 if args.workload == "eval" else 32
    return args


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_or_join(args):
    """Returns (rank, world, local_rank) of this process -- after making sure
    that `world` really is args.gpus.  A bare `python bench.py --gpus N` (N > 1)
    re-executes under torch.distributed.run and never returns."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None:
        if args.gpus > 1:
            have = torch.cuda.device_count()
            if have < args.gpus:
                sys.exit("bench.py: --gpus %d requested but only %d GPU(s) visible; refusing to report a "
                         "%d-GPU number from fewer devices" % (args.gpus, have, args.gpus))
            env = dict(os.environ)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                   "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
                   "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
            sys.exit(subprocess.call(cmd, env=env))
        return 0, 1, 0
    world = int(env_world)
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d)"
                 % (args.gpus, world, args.gpus))
    return int(os.environ.get("RANK", "0")), world, int(os.environ.get("LOCAL_RANK", "0"))


def init_ranks(world, local_rank):
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if torch.cuda.device_count() <= local_rank:
        sys.exit("bench.py: local rank %d has no GPU (%d visible)" % (local_rank, torch.cuda.device_count()))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
        assert dist.get_world_size() == world
        # every rank must own a different device (RCCL-visible rank count == GPUs in use)
        ids = [None] * world
        dist.all_gather_object(ids, torch.cuda.current_device())
        assert len(set(ids)) == world, "ranks share a device: %r" % (ids,)
    return dev


def eval_step(cd_mod, emd_mod, fscore, pred, gt, eps, iters, ev=None):
    """completion/model_utils.py:67-85 (calc_cd with calc_f1, calc_emd)."""
    if ev:
        ev[0].record()
    dist1, dist2, _, _ = cd_mod(gt, pred)
    cd_p = (torch.sqrt(dist1).mean(1) + torch.sqrt(dist2).mean(1)) / 2
    cd_t = dist1.mean(1) + dist2.mean(1)
    f1, _, _ = fscore(dist1, dist2)
    if ev:
        ev[1].record()
    dist, _ = emd_mod(pred, gt, eps, iters)
    emd_out = torch.sqrt(dist).mean(1)
    if ev:
        ev[2].record()
    return cd_p, cd_t, f1, emd_out


def cpu_baseline(args, n):
    """The CPU oracle (C restatement of the reference kernels, OpenMP over the
    batch dimension) on a bounded sample of the same workload."""
    import numpy as np
    import oracle  # test infrastructure: timed here as the baseline, never shipped
    oracle.build()
    cores = os.cpu_count() or 1
    oracle.set_num_threads(cores)
    bs = args.cpu_sample or min(cores, 64)
    rng = np.random.default_rng(0)
    pred = rng.random((bs, n, 3), dtype=np.float32)
    gt = rng.random((bs, n, 3), dtype=np.float32)
    t0 = time.perf_counter()
    oracle.chamfer_forward(gt, pred)
    t1 = time.perf_counter()
    _, _, stats = oracle.emd_forward(pred, gt, args.eps, args.iters, return_stats=True)
    t2 = time.perf_counter()
    return {
        "value": bs * n * n / (t2 - t0),
        "unit": "point-pairs/s",
        "cores": min(cores, bs),
        "kind": "port",
        "sample": "%d clouds of %d pts (one per core, OpenMP over clouds), CD + EMD eps=%g iters=%d; "
                  "CD %.2f s, EMD %.2f s, EMD rounds %d, bids/cloud %.0f" % (
                      bs, n, args.eps, args.iters, t1 - t0, t2 - t1,
                      int(stats[:, 0].max()), float(stats[:, 1].mean())),
    }


def gpu_reference_baseline(args, pred, gt, our_cd_ms, our_emd_ms):
    """The REFERENCE's own Chamfer and EMD kernels on the SAME resident batch of this GPU: utils/metrics/CD/chamfer3D and
    utils/metrics/EMD compiled for gfx950 AS WRITTEN by oracle/build_ref_gpu.sh (oracle/_ref: test infrastructure, built
    where the reference tree is present and shipped as .so files).  A baseline beside `cpu_baseline`, measured after the
    timed region; the product never loads it.  None when oracle/_ref is not there."""
    try:
        from oracle import ref_gpu
        if not ref_gpu.available(""):
            return None
        ref_gpu.chamfer_forward(gt, pred)                       # warm-up: module load, first launch
        t0 = time.perf_counter()
        for _ in range(3):
            ref_gpu.chamfer_forward(gt, pred)                   # (each call allocates its outputs and synchronises)
        cd_ms = (time.perf_counter() - t0) / 3 * 1e3
        t0 = time.perf_counter()
        for _ in range(2):
            d, _, _ = ref_gpu.emd_forward(pred, gt, args.eps, args.iters)
        emd_ms = (time.perf_counter() - t0) / 2 * 1e3
    except Exception as e:   # a baseline must not break the bench line
        return {"error": "%s: %s" % (type(e).__name__, e)}
    B, n = pred.shape[0], pred.shape[1]
    return {"value": float(B) * n * n / ((cd_ms + emd_ms) * 1e-3), "unit": "point-pairs/s", "kind": "reference kernels on this GPU",
            "cd_ms": cd_ms, "emd_ms": emd_ms, "ms_per_step": cd_ms + emd_ms,
            "this_repo_ms": {"cd_f1_ms": our_cd_ms, "emd_ms": our_emd_ms},
            "speedup_of_this_repo": (cd_ms + emd_ms) / (our_cd_ms + our_emd_ms),
            "emd_mean_sqrt_dist": float(d.sqrt().mean()),
            "sample": "the timed batch itself: chamfer_cuda_forward + emd_cuda_forward (eps=%g, iters=%d) on (%d,%d,3), wrapper "
                      "allocations included, F-score (Python on both sides) not; chamfer3D.cu / emd_cuda.cu through hipify-perl + "
                      "hipcc --offload-arch=gfx950 with default flags (DESIGN 2.2)" % (args.eps, args.iters, B, n)}


def cpu_reference_path(n):
    """The reference's only CPU path on this op layer: distChamfer
    (utils/metrics/CD/chamfer_python.py:18-39: float64, |x|^2+|y|^2-2 bmm, two
    full-matrix mins), in-tree restatement (bit-identical to the imported
    reference on the golden fixtures, tests/test_host.py), all torch threads.
    cfg 1 = (4, 2048, 2048) whole batch, best of 5; headline CD shape =
    (., n, n) one cloud per chunk, a bounded sample of clouds."""
    from mvp_benchmark_amd.metrics.CD.chamfer_python import distChamfer
    cores = os.cpu_count() or 1
    g = torch.Generator().manual_seed(0)
    a, b = torch.rand(4, 2048, 3, generator=g), torch.rand(4, 2048, 3, generator=g)
    clouds = 2
    a2, b2 = torch.rand(clouds, n, 3, generator=g), torch.rand(clouds, n, 3, generator=g)
    out = {"kind": "reference (in-tree restatement of distChamfer, fp64 bmm)", "cores": cores,
           "unit": "point-pairs/s", "by_torch_threads": {}}
    # all host threads as BASELINE.md says -- and fewer: a (4, 2048, 2048) fp64 bmm + two mins does
    # not scale to hundreds of threads, the best setting is the fair baseline
    for threads in sorted({cores, min(cores, 32), min(cores, 8)}, reverse=True):
        torch.set_num_threads(threads)
        distChamfer(a, b)
        best = min(_timeit(lambda: distChamfer(a, b)) for _ in range(5))
        t2 = _timeit(lambda: distChamfer(a2, b2, chunk=1))
        out["by_torch_threads"][str(threads)] = {
            "cfg1_4x2048x2048": {"seconds": best, "value": 4 * 2048 * 2048 / best},
            "headline_cd_%dx%d" % (n, n): {"seconds": t2, "clouds": clouds, "chunk": 1,
                                           "value": clouds * float(n) * n / t2}}
    for key in ("cfg1_4x2048x2048", "headline_cd_%dx%d" % (n, n)):
        th, rec = max(out["by_torch_threads"].items(), key=lambda kv: kv[1][key]["value"])
        out[key] = dict(rec[key], torch_threads=int(th))
    return out


def _timeit(fn):
    t0 = time.perf_counter()
    fn()
    return time.perf_counter() - t0


def settle(fn, max_steps=200, window=3, tol=0.05):
    """Untimed pre-warm-up of a MODEL step: MIOpen serves a convolution shape from its naive reference kernels
    (naive_conv_*: 17 ms per call in PCN's forward) for the first calls of a process, for how many differs from box to box
    (profiles/r6d_pcn_eval_kernel_stats.txt; the same step measured 22.9 and 51.7 ms with one warm-up step).  Runs the step
    until `window` consecutive steps are within `tol` of their minimum (or max_steps); returns the steps it took."""
    sync = torch.cuda.synchronize if torch.cuda.is_available() else (lambda: None)
    times = []
    for i in range(max_steps):
        sync()
        t0 = time.perf_counter()
        fn()
        sync()
        times.append(time.perf_counter() - t0)
        if i + 1 >= window + 2:   # (the first two steps carry allocations and the solver search itself)
            w = times[-window:]
            if max(w) <= (1.0 + tol) * min(w) and min(w) <= (1.0 + tol) * min(times):
                return i + 1
    return max_steps


def emd_sources_digest():
    """sha256 over the EMD kernels' sources: what a set of committed counters was taken on (tools/make_traffic.py
    stamps profiles/traffic.json with it)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "mvp_benchmark_amd", "csrc", "emd*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def committed_counters(B, n, eps, iters):
    """Per-call counters of mvp_emd_forward (its kernels summed) from the committed rocprofv3
    --pmc passes (profiles/traffic.json), only if they were taken on this shape.  `counters_current` says
    whether the EMD sources are still the ones the counters were taken on (the file carries their digest)."""
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["emd_forward"]
        if (tr["batch"], tr["points"], tr["eps"], tr["iters"]) == (B, n, eps, iters):
            tr = dict(tr)
            tr["counters_current"] = tr.get("emd_sources_sha256") == emd_sources_digest()
            return tr
    except (OSError, KeyError, ValueError):
        pass
    return None


def side_measurements(args, dev, g, emd_mod, furthest_point_sample, gather_points, reps=3):
    """Untimed side measurements on rank 0, AFTER the timed region: the EMD sweep of BASELINE
    cfg 4 (B = 64, n in {1024, 2048, 4096, 8192}, eval setting), cfg 2's EMD (B = 32 at the headline
    size), the training setting (eps 0.005, 50 rounds), the auction on surface-shaped clouds (sphere / chair: gt + noise
    0.03 and independent samples, at the headline size and at 2048 points) and FPS at two sizes.  Every job runs once
    untimed and then `reps` times; the figure is the mean (about 2.5 s of GPU time in all)."""
    B, n = args.batch, args.points
    jobs = []
    for (jb, jn, jeps, jit, key) in ((B, 1024, args.eps, args.iters, "emd_cfg4_n1024_ms"),
                                     (B, 2048, args.eps, args.iters, "emd_cfg4_n2048_ms"),
                                     (B, 4096, args.eps, args.iters, "emd_cfg4_n4096_ms"),
                                     (B, 8192, args.eps, args.iters, "emd_cfg4_n8192_ms"),
                                     (max(1, B // 2), n, args.eps, args.iters, "emd_cfg2_b%d_n%d_ms" % (max(1, B // 2), n)),
                                     (B, n, 0.005, 50, "emd_train_setting_n%d_ms" % n)):
        a = torch.rand(jb, jn, 3, generator=g).to(dev)
        b_ = torch.rand(jb, jn, 3, generator=g).to(dev)
        jobs.append((key, (lambda a=a, b_=b_, jeps=jeps, jit=jit: emd_mod(a, b_, jeps, jit))))
    # The auction on what the eval loop really sees (completion/dataset.py:21-34: MVP clouds are samples of surfaces; a
    # trained network's prediction is its target plus noise): gt + noise 0.03 and two independent samples of a sphere and of
    # a chair-like shape, at the headline size and at the shipped cfgs' 2048 points.  Same call, same setting.
    from mvp_benchmark_amd.synthetic import prediction_pair
    for shape in ("sphere", "chair"):
        for mode, tag in (("0.03", "gt_plus_noise_0.03"), ("indep", "independent_samples")):
            for jn in (n, 2048):
                p_, g_ = prediction_pair(shape, mode, g, B, jn)
                p_, g_ = p_.to(dev), g_.to(dev)
                jobs.append(("emd_surface_%s_%s_n%d_ms" % (shape, tag, jn),
                             (lambda p_=p_, g_=g_: emd_mod(p_, g_, args.eps, args.iters))))
    fps_meta = {}
    for (fn, fm) in ((n, 2048), (2048, 512)):
        x = torch.rand(B, fn, 3, generator=g).to(dev)
        key = "fps_%d_to_%d" % (fn, fm)
        fps_meta[key] = (fn, fm)
        jobs.append((key, (lambda x=x, fm=fm: furthest_point_sample(x, fm))))
    for _, fn_ in jobs:          # warm-up (allocator, first-launch costs)
        fn_()
    torch.cuda.synchronize()
    tot = {k: [0.0, 0] for k, _ in jobs}
    passes = 0
    while passes < reps:
        for k, fn_ in jobs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn_()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            tot[k][0] += ms
            tot[k][1] += 1
        passes += 1
    out = {"side_measurement_reps": passes}
    for k, (ms_sum, cnt) in tot.items():
        ms = ms_sum / cnt
        if k in fps_meta:
            fn, fm = fps_meta[k]
            out[k] = {"ms": ms, "sampled_pts_per_s": B * fm / ms * 1e3, "point_updates_per_s": B * (fm - 1) * fn / ms * 1e3}
        else:
            out[k] = ms
    # one gather through the sampled indices (keeps the C3 -> E1 chain of the models exercised)
    x = torch.rand(B, 2048, 3, generator=g).to(dev)
    gather_points(x.transpose(1, 2).contiguous(), furthest_point_sample(x, 512))
    torch.cuda.synchronize()
    return out


def model_level_note(args):
    return ("5 timed steps each after the timed region, after an untimed pre-warm-up until three consecutive steps agree within "
            "5 %% (MIOpen's first calls of a shape run its naive kernels); vrcnet: 32 x 2048 points per rank, fused Adam; "
            "pcn_eval: 32 clouds, forward 2048 -> %d + CD/F1 of the output + EMD eps %g x %d on chair gt + noise 0.03"
            % (args.points, args.eps, args.iters))


def model_level_measurements(args, dev, with_reference):
    """BASELINE cfgs 3 and 2 in front of the driver, rank 0, AFTER the timed region, 5 timed steps each:
      vrcnet_train_ms  cfg 3's per-rank step (completion/train.py:122-142: forward, CD losses + KLD, backward, fused Adam),
                       32 clouds of 2048 points, cfgs/vrcnet.yaml, random weights;
      pcn_eval_ms      cfg 2's eval step (completion/models/pcn.py:105-112 + model_utils.calc_emd): PCN forward 2048 ->
                       16384 + CD / F1 of its output + EMD (eps 0.004, 3000 rounds) -- the EMD on gt + noise 0.03 of a
                       chair-like surface, what a TRAINED network's output looks like next to its target (a random-init PCN
                       emits one blob: bench.py --workload pcn_eval).
    with_reference (oracle/_ref travelled; the baseline leg, tests/report_reference_model_step.py's method): the same steps in
    the reference's formulation on the reference's own operator kernels compiled for this GPU."""
    import importlib
    sys.path.insert(0, os.path.join(ROOT, "completion"))
    import model_utils as mu
    import op_config
    import train
    import mvp_benchmark_amd.pointwise as pw
    from mvp_benchmark_amd.synthetic import prediction_pair
    out = {}

    def timed(fn, warm, reps=5, key=None):
        for _ in range(warm):
            fn()
        if key:
            out[key + "_settled_after_steps"] = settle(fn)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    g = torch.Generator().manual_seed(4242)
    gt2k = torch.rand(32, 2048, 3, generator=g).to(dev)
    part2k = torch.rand(32, 2048, 3, generator=g).to(dev).transpose(2, 1).contiguous()
    surf_pred, surf_gt = [t.to(dev) for t in prediction_pair("chair", "0.03", g, 32, args.points)]
    part16 = torch.rand(32, 3, 2048, generator=g).to(dev)

    def vrcnet_step_fn(fused, switches=None):
        cfg = train.load_config(os.path.join(ROOT, "completion", "cfgs", "vrcnet.yaml"))   # (sets the op-layer switches)
        cfg.load_model = None
        if switches:
            op_config.configure(**switches)
        torch.manual_seed(0)
        net = importlib.import_module("models.vrcnet").Model(cfg).to(dev).train()
        opt = torch.optim.Adam(net.parameters(), lr=cfg.lr, betas=(0.9, 0.999), fused=fused)

        def step():
            opt.zero_grad()
            _, _, loss = net(part2k, gt2k, alpha=0.5)
            loss.mean().backward()
            opt.step()
        return step

    def pcn_step_fn(switches=None):
        cfg = train.load_config(os.path.join(ROOT, "completion", "cfgs", "pcn_eval16k.yaml"))
        cfg.eval_emd = False
        if switches:
            op_config.configure(**switches)
        torch.manual_seed(1)
        net = importlib.import_module("models.pcn").Model(cfg).to(dev).eval()

        def step():
            with torch.no_grad():
                net(part16, surf_gt, prefix="val")                                      # forward + calc_cd(out2, gt, calc_f1=True)
                mu.calc_emd(surf_pred, surf_gt, eps=args.eps, iterations=args.iters)
        return step

    out["vrcnet_train_ms"] = timed(vrcnet_step_fn(True), 2, key="vrcnet_train")
    out["vrcnet_train_samples_per_s"] = 32e3 / out["vrcnet_train_ms"]
    out["pcn_eval_ms"] = timed(pcn_step_fn(), 1, key="pcn_eval")
    out["pcn_eval_clouds_per_s"] = 32e3 / out["pcn_eval_ms"]
    mu.check_emd_status()
    out["model_level_note"] = model_level_note(args)
    if with_reference:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import ref_ops                      # the reference's own kernels behind the operator API (test infrastructure)
            from models import _common, ecg, edge_unet, pcn, relational, vrcnet
            off = dict(gather_sum=0, gather_max=0, side_lanes=0, stacked_projections=0, skip_full_fps_of_gt=0,
                       conv_before_interp=0, folded_conv=0)
            pw.MFMA_TRAIN = pw.USE_MFMA = False
            undo = ref_ops.patch_ops([mu, _common, relational, edge_unet, ecg, vrcnet, pcn])
            ref_ops.ref.SYNC = False            # like the reference's wrappers: no host synchronisation per operator
            try:
                out["vrcnet_train_reference_kernels_ms"] = timed(vrcnet_step_fn(False, off), 1, reps=3)
                out["pcn_eval_reference_kernels_ms"] = timed(pcn_step_fn(off), 1, reps=2)
            finally:
                ref_ops.ref.SYNC = True
                undo()
                op_config.OPS.reset()
                pw.MFMA_TRAIN = pw.USE_MFMA = True
        except Exception as exc:   # (the baseline must never cost the measurement)
            out["model_level_reference_error"] = repr(exc)[:200]
    return out


def run_eval(args, rank, world, dev):
    from mvp_benchmark_amd.metrics import cd, emd, fscore
    from mvp_benchmark_amd.mm3d_pn2 import furthest_point_sample, gather_points
    from mvp_benchmark_amd import _lib

    B, n = args.batch, args.points
    g = torch.Generator().manual_seed(1000 + rank)
    pred = torch.rand(B, n, 3, generator=g).to(dev)
    gt = torch.rand(B, n, 3, generator=g).to(dev)
    cd_mod, emd_mod = cd(), emd()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eval_step(cd_mod, emd_mod, fscore, pred, gt, args.eps, args.iters)
    barrier()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for s in range(args.steps):
        out = eval_step(cd_mod, emd_mod, fscore, pred, gt, args.eps, args.iters, evs[s])
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    # eval-loop aggregation: one 5-float sum all-reduce
    sums = torch.stack([out[0].sum(), out[1].sum(), out[2].sum(), out[3].sum(),
                        torch.tensor(float(B), device=dev)]).double()
    if world > 1:
        torch.distributed.all_reduce(sums)
    sums = sums.cpu()

    cd_ms = sum(e[0].elapsed_time(e[1]) for e in evs) / args.steps
    emd_ms = sum(e[1].elapsed_time(e[2]) for e in evs) / args.steps
    if rank != 0:
        return None

    # ---- rank 0 only: side measurements (outside the timed region) ----
    # auction statistics of one EMD launch (rounds, bids) via the C ABI
    nbytes = _lib.emd_scratch_bytes(B, n)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dist = torch.zeros(B, n, device=dev)
    ass = torch.zeros(B, n, dtype=torch.int32, device=dev)
    _lib.call("mvp_emd_forward", dev, B, n, pred, gt, dist, ass, args.eps, args.iters, scratch, nbytes)
    torch.cuda.synchronize()
    stats = scratch[nbytes - B * 16:].view(torch.int64).view(B, 2).cpu()
    rounds, bids = int(stats[:, 0].max()), float(stats[:, 1].double().mean())

    # the hand-over records (per cloud, right before the statistics; _lib.emd_records): round at which the
    # lean kernel took the cloud over (0: never), persons unassigned at the LAST hand-over (round 300
    # when the cluster widths are dealt out again), cluster width of the launch that finished the cloud
    rec = _lib.emd_records(scratch, nbytes, B)
    handover = {"round_min": int(rec["first_handover"].min()), "round_max": int(rec["first_handover"].max()),
                "unassigned_max_at_last_handover": int(rec["unassigned"].max()),
                "clouds_by_final_width": {str(int(w)): int((rec["final_width"] == w).sum()) for w in sorted(set(rec["final_width"].tolist()))}}

    # (N > 1: the other ranks are already waiting to leave -- two passes only)
    side = {} if args.no_side else side_measurements(args, dev, g, emd_mod, furthest_point_sample, gather_points,
                                                      reps=3 if world == 1 else 1)

    pairs = float(B) * n * n
    value = pairs * world / (elapsed / args.steps)
    emd_bytes = 32.0 * B * n  # xyz1+xyz2 in (24 B/pt) + dist+assignment out (8 B/pt)
    achieved = emd_bytes / (emd_ms * 1e-3) / 1e9
    ctr = committed_counters(B, n, args.eps, args.iters)
    traffic = (ctr["FETCH_SIZE_KB"] + ctr["WRITE_SIZE_KB"]) * 1024.0 if ctr else None
    # second roof: the kernel is a chain of dependent bids, not a stream.  VALU issue
    # fraction = wave-level VALU instructions per launch (PMC, committed) x 64 lanes /
    # (kernel time x the chip's lane-op rate); instructions per bid from the same pass.
    issue = None
    if ctr and "SQ_INSTS_VALU" in ctr:
        total_bids = bids * B
        issue = {"valu_insts_per_launch": ctr["SQ_INSTS_VALU"], "salu_insts_per_launch": ctr.get("SQ_INSTS_SALU"),
                 "valu_insts_per_bid": ctr["SQ_INSTS_VALU"] / total_bids,
                 "salu_insts_per_bid": (ctr.get("SQ_INSTS_SALU") or 0) / total_bids,
                 "valu_issue_frac": ctr["SQ_INSTS_VALU"] * 64 / (emd_ms * 1e-3 * VALU_LANE_OPS_PER_S),
                 "wait_any_frac": ctr.get("wait_any_frac"), "source": ctr.get("source"),
                 # False: the kernels changed since the committed passes (figures of an older build, kept for the trend)
                 "counters_current": ctr.get("counters_current"), "counters_taken_at": ctr.get("taken_at")}
    line = {
        "metric": "point-pairs/sec CD+EMD @2048->16384 pts, batch 64",
        "value": value,
        "unit": "point-pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "completion eval CD+F1+EMD, pred/gt (%d,%d,3) per GPU, "
                               "EMD eps=%g iters=%d" % (B, n, args.eps, args.iters),
                   "batch_per_gpu": B, "points": n, "parallelism": "batch-sharded x%d" % world},
        # the HBM figures are what the contract asks for; the kernel itself is bound by the
        # latency of its dependent bid chain, hence bound = "latency" and the
        # issue-side roof next to it
        # kernel: one mvp_emd_forward call = emd_auction_kernel (the rounds with four bidders per wave,
        # ~100 of 3000), emd_lean_kernel (to round 300) and emd_lean_tiers_kernel (the rest, cluster
        # widths by load); the events bracket the call, the rocprofv3 kernel trace lists the three
        # durations, whose sum must agree
        "roofline": {"kernel": "emd_auction_kernel + emd_lean_kernel + emd_lean_tiers_kernel (one mvp_emd_forward)", "bound": "latency", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "issue": issue,
                     "us_per_round": emd_ms * 1e3 / max(rounds, 1)},
        "extra": {
            "cd_f1_ms": cd_ms, "emd_ms": emd_ms,
            "emd_rounds_max": rounds, "emd_bids_per_cloud": bids,
            "emd_reference_pair_evals_per_s": bids * n * B / (emd_ms * 1e-3),
            # brute-force-EQUIVALENT rates (all B*N*M pairs / time): the Morton-sorted kernel
            # evaluates only ~9 % of them, so these are not hardware rates
            "cd_bruteforce_equivalent_pairs_per_s": 2 * pairs / (cd_ms * 1e-3),
            "cd_hbm_GBs_20B_per_point": 20.0 * B * 2 * n / (cd_ms * 1e-3) / 1e9,
            "metrics": {"cd_p": float(sums[0] / sums[4]), "cd_t": float(sums[1] / sums[4]),
                        "f1": float(sums[2] / sums[4]), "emd": float(sums[3] / sums[4])},
            "emd_handover": handover,
            **side,
        },
    }
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args, n)
        line["cpu_baseline_reference_path"] = cpu_reference_path(n)
        line["gpu_reference_baseline"] = gpu_reference_baseline(args, pred, gt, cd_ms, emd_ms)
    if world == 1 and not args.no_side:
        # cfg 3's per-rank train step and cfg 2's eval step (after everything timed above; the reference-kernel leg only
        # with the baselines)
        ref_ok = not args.no_cpu_baseline and isinstance(line.get("gpu_reference_baseline"), dict) and "error" not in line["gpu_reference_baseline"]
        del pred, gt
        torch.cuda.empty_cache()
        try:
            line["extra"].update(model_level_measurements(args, dev, ref_ok))
        except Exception as exc:
            line["extra"]["model_level_error"] = repr(exc)[:300]
    return line


def run_vrcnet_train(args, rank, world, dev):
    """BASELINE cfg 3: VRCNet DDP training, batch 32 per rank (256 on 8 GPUs)."""
    sys.path.insert(0, os.path.join(ROOT, "completion"))
    import importlib
    import train
    from train_utils import unwrap

    cfg = train.load_config(os.path.join(ROOT, "completion", "cfgs", "vrcnet.yaml"))
    cfg.load_model = None
    B = args.batch
    torch.manual_seed(0)                      # identical initial weights on every rank
    net = importlib.import_module("models.vrcnet").Model(cfg).to(dev)
    net = train.wrap_ddp(net, dev, world, cfg)
    torch.manual_seed(1000 * (rank + 1))      # per-rank dropout / rsample streams
    opt = torch.optim.Adam(unwrap(net).parameters(), lr=cfg.lr, betas=(0.9, 0.999), fused=True)   # (as completion/train.py builds it)
    g = torch.Generator().manual_seed(1000 + rank)
    gt = torch.rand(B, cfg.num_points, 3, generator=g).to(dev)
    partial = torch.rand(B, 2048, 3, generator=g).to(dev).transpose(2, 1).contiguous()
    scale = train.loss_scale(cfg, world)

    def step():
        opt.zero_grad()
        _, _, loss = net(partial, gt, alpha=0.5)
        (loss.mean() * scale).backward()
        opt.step()
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    # gradient all-reduce volume = every parameter that receives a gradient, fp32
    grad_bytes = sum(p.numel() for p in unwrap(net).parameters() if p.grad is not None) * 4
    bus = None
    if world > 1:
        buf = torch.empty(grad_bytes // 4, device=dev)
        for _ in range(2):
            torch.distributed.all_reduce(buf)
        barrier()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            torch.distributed.all_reduce(buf)
        barrier()
        ar = (time.perf_counter() - t0) / reps
        bus = {"allreduce_ms": ar * 1e3, "algbw_GBs": grad_bytes / ar / 1e9,
               "busbw_GBs": grad_bytes / ar / 1e9 * 2 * (world - 1) / world}
    if rank != 0:
        return None
    ms = elapsed / args.steps * 1e3
    # The step's dominant kernels are the float32-MFMA GEMMs of the 1x1 convolutions (csrc/pointwise_mfma.hip,
    # ~10 of the step's ms): their three passes at the step's largest layer, 2 B x (512 -> 1024) x 2048, timed with
    # events on the current stream (after the timed region, 5 repetitions), against the f32 MFMA peak.
    from mvp_benchmark_amd.pointwise import mfma_linear, mfma_wgrad
    cb, cin, cout, L = 2 * B, 512, 1024, 2048
    gx = torch.Generator().manual_seed(5)
    x = torch.randn(cb, cin, L, generator=gx).to(dev)
    w = torch.randn(cout, cin, generator=gx).to(dev)
    gy = torch.randn(cb, cout, L, generator=gx).to(dev)
    flops = 2.0 * cb * cin * cout * L

    def tflops(fn, reps=5):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return flops / (e0.elapsed_time(e1) / reps * 1e-3) / 1e12

    passes = {"forward": tflops(lambda: mfma_linear(x, w)), "data_gradient": tflops(lambda: mfma_linear(gy, w, w_kmajor=True)),
              "weight_gradient": tflops(lambda: mfma_wgrad(x, gy, cout, cin, True))}
    achieved = 3.0 / sum(1.0 / v for v in passes.values())          # the three passes back to back
    roofline = {"kernel": "pointwise_mfma_kernel (forward, data gradient) + pointwise_wgrad_mfma_kernel, (%d, 512 -> 1024, 2048)" % cb,
                "bound": "mfma", "achieved": achieved, "peak": 157.3, "unit": "TFLOP/s", "frac": achieved / 157.3,
                "traffic": None, "per_pass_tflops": passes,
                "note": "float32 in / float32 accumulate (v_mfma_f32_32x32x2_f32); peak at 2.4 GHz -- the chip holds ~2.05 GHz "
                        "under this load (profiles/r4_pmc_pointwise_512_1024.json)"}
    return {
        "metric": "VRCNet train samples/sec (cfgs/vrcnet.yaml, DDP, batch 32 per GPU)",
        "value": B * world / (ms * 1e-3),
        "unit": "samples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms,
        "steps_per_s": 1e3 / ms,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "VRCNet DDP train step (forward, CD losses + KLD, backward, Adam), "
                               "%d clouds of 2048 pts per GPU" % B,
                   "global_batch": B * world, "batch_per_gpu": B, "parallelism": "ddp x%d" % world},
        "grad_allreduce_bytes_per_step": grad_bytes,
        "rccl": bus,
        "final_loss": float(loss.mean()),
        "roofline": roofline,
    }


def run_pcn_eval(args, rank, world, dev):
    """BASELINE cfg 2: the completion eval step of PCN, 2048 -> 16384 points, 32 clouds per rank:
    net(partial, gt, prefix="val") = forward + calc_cd(out2, gt, calc_f1=True) (models/pcn.py:75-112,
    model_utils.py:35-49), then calc_emd(pred, gt, eps 0.004, 3000 rounds) (model_utils.py:51-57), as val() /
    test.py run them per batch.  Weights are random (no checkpoint exists here): a random-init PCN emits one tight
    blob, and an auction between a blob and a spread cloud is a degenerate input that says nothing about the
    eval loop of a trained network (orders of magnitude more bids; measured once below, outside the timed
    region).  The timed EMD therefore runs on what a TRAINED network's prediction looks like next to its target:
    ground truth + noise 0.03 on a chair-like surface (mvp_benchmark_amd/synthetic.py; VERDICT r5: the figure with a
    uniform stand-in flattered the step then -- it is kept beside it in `extra`); the network's forward pass and the
    CD / F1 of its real output are timed as they are."""
    sys.path.insert(0, os.path.join(ROOT, "completion"))
    import importlib
    import model_utils as mu
    import train

    cfg = train.load_config(os.path.join(ROOT, "completion", "cfgs", "pcn_eval16k.yaml"))
    cfg.eval_emd = False              # (EMD is called beside the model below, on the spread stand-in)
    B, n = args.batch, int(cfg.num_points)
    torch.manual_seed(1)
    net = importlib.import_module("models.pcn").Model(cfg).to(dev).eval()
    g = torch.Generator().manual_seed(1000 + rank)
    partial = torch.rand(B, 3, 2048, generator=g).to(dev)
    gt = torch.rand(B, n, 3, generator=g).to(dev)
    pred_like = torch.rand(B, n, 3, generator=g).to(dev)
    # what a TRAINED network's prediction looks like next to its target: gt + noise 0.03 on a chair-like surface (the
    # timed EMD; the independent uniform stand-in of rounds 3-5 is timed beside it: extra)
    from mvp_benchmark_amd.synthetic import prediction_pair
    surf_pred, surf_gt = [t.to(dev) for t in prediction_pair("chair", "0.03", g, B, n)]

    def step():
        with torch.no_grad():
            r = net(partial, gt, prefix="val")
            e = mu.calc_emd(surf_pred, surf_gt, eps=args.eps, iterations=args.iters)
        return r, e

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    settled = settle(step)   # (untimed, before the W warm-up steps: MIOpen's naive first calls, see settle)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r, e = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return None

    # per-part milliseconds, outside the timed region (events on the current stream, mean of 5)
    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, out

    with torch.no_grad():
        fwd_ms, out = timed(lambda: net(partial, prefix="test"))
        out2 = out["result"]
        cd_ms, _ = timed(lambda: mu.calc_cd(out2, gt, calc_f1=True))
        emd_ms, _ = timed(lambda: mu.calc_emd(pred_like, gt, eps=args.eps, iterations=args.iters))
        surf_ms, _ = timed(lambda: mu.calc_emd(surf_pred, surf_gt, eps=args.eps, iterations=args.iters), reps=3)
        blob_ms, _ = timed(lambda: mu.calc_emd(out2, gt, eps=args.eps, iterations=args.iters), reps=1)
    mu.check_emd_status()
    ms = elapsed / args.steps * 1e3
    return {
        "metric": "PCN completion eval clouds/sec (cfgs/pcn_eval16k.yaml: 2048 -> 16384 pts, forward + CD + F1 on the network's "
                  "output + EMD on a stand-in prediction (chair-like surface, ground truth + noise 0.03), batch 32 per GPU)",
        "value": B * world / (ms * 1e-3),
        "unit": "clouds/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms,
        "steps_per_s": 1e3 / ms,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (random-init PCN weights, uniform clouds; EMD on a surface-shaped pair: prediction = ground truth + noise 0.03)",
        "config": {"workload": "PCN eval step: forward (2048 -> %d pts) + calc_cd(f1) on its output + calc_emd eps=%g iters=%d, "
                               "%d clouds per GPU" % (n, args.eps, args.iters, B),
                   "batch_per_gpu": B, "points": n, "parallelism": "batch-sharded x%d" % world},
        "parts_ms": {"pcn_forward": fwd_ms, "calc_cd_f1_on_network_output": cd_ms, "calc_emd_chair_gt_plus_noise_0.03": surf_ms,
                     "sum": fwd_ms + cd_ms + surf_ms},
        "extra": {"settled_after_steps": settled, "calc_emd_independent_uniform_stand_in_ms": emd_ms,
                  "clouds_per_s_with_emd_on_independent_uniform_stand_in": B * world / ((fwd_ms + cd_ms + emd_ms) * 1e-3),
                  "calc_emd_on_random_init_output_ms": blob_ms,
                  "note": "random-init PCN output is one tight blob: the auction against a spread cloud is the degenerate case "
                          "(every person bids every round); shown for completeness, not part of the timed step",
                  "metrics": {k: float(r[k].mean()) for k in ("cd_p", "cd_t", "f1")}, "emd_mean": float(e.mean())},
    }


def main():
    args = parse()
    rank, world, local_rank = launch_or_join(args)
    dev = init_ranks(world, local_rank)
    runner = {"eval": run_eval, "vrcnet_train": run_vrcnet_train, "pcn_eval": run_pcn_eval}[args.workload]
    line = runner(args, rank, world, dev)
    if line is not None:
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
