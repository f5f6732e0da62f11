"""BASELINE cfg 5: DCP registration, 128 pairs of 1024 points, one MI355X -- forward (eval) and forward +
backward (training step without the optimizer) timings, with the op-layer split from the torch profiler
(kNN graph, neighbour gather / its gradient, the 3x3 Kabsch SVD launch).
   python tools/bench_dcp.py            (MVP_BENCH_REPS: timed repetitions)"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = os.path.join(ROOT, "registration")
sys.path.insert(0, ROOT); sys.path.insert(0, REG); sys.path.insert(0, os.path.join(REG, "models"))
import torch
import dcp
import train_utils as tu

REPS = int(os.environ.get("MVP_BENCH_REPS", "10"))
dev = "cuda:0"
B, N = 128, 1024
torch.manual_seed(0)
net = dcp.Model(types.SimpleNamespace()).to(dev)
g = torch.Generator().manual_seed(5)
src = (torch.rand(B, N, 3, generator=g) - 0.5).to(dev)
q = torch.nn.functional.normalize(torch.randn(B, 4, generator=g), dim=1).to(dev)
Rg = tu.quat2mat(q)
tg = (torch.rand(B, 3, generator=g) - 0.5).to(dev)
tgt = src @ Rg.transpose(1, 2) + tg.unsqueeze(1)
T_gt = tu.rt_to_transformation(Rg, tg.unsqueeze(2))


def timed(fn, reps=REPS):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def fwd():
    with torch.no_grad():
        return net(src, tgt, T_gt)


def fwd_bwd():
    net.zero_grad(set_to_none=True)
    loss = net(src, tgt, T_gt)[0]
    loss.mean().backward()


net.eval()
ms_f = timed(fwd)
net.train()
ms_fb = timed(fwd_bwd)
print("cfg 5 DCP (%d pairs x %d points, k = 20 graph, 5.57 M parameters): forward %.2f ms (%.0f pairs/s), "
      "forward + backward %.2f ms (%.0f pairs/s)" % (B, N, ms_f, B / ms_f * 1e3, ms_fb, B / ms_fb * 1e3), flush=True)

from torch.profiler import profile, ProfilerActivity
for name, fn in (("forward", fwd), ("forward + backward", fwd_bwd)):
    (net.eval() if name == "forward" else net.train())
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
    # kernel rows only (operator rows carry their kernels' time a second time)
    rows = [(e.key, e.self_device_time_total / 3e3, e.count // 3) for e in prof.key_averages()
            if e.self_device_time_total > 0 and str(e.device_type).endswith("CUDA")]
    total = sum(r[1] for r in rows)
    def share(*subs):
        return sum(r[1] for r in rows if any(s in r[0] for s in subs))
    knn = share("knn_kernel")
    gather = share("gather_lds_kernel", "gather_kernel", "group_points")
    scatter = share("scatter_lds_kernel", "transposed_reduce_kernel", "transpose_index_kernel", "_grad_kernel")
    svd = share("svd3")
    print("  %s: GPU time %.2f ms per step = kNN graph %.2f + neighbour gather %.2f + gather gradient %.2f + "
          "Kabsch SVD3 %.3f + everything else (library GEMMs / attention / elementwise) %.2f" % (
              name, total, knn, gather, scatter, svd, total - knn - gather - scatter - svd), flush=True)
    for r in sorted(rows, key=lambda r: -r[1])[:8]:
        print("      %-90s %8.3f ms x%d" % (r[0][:90], r[1], r[2]), flush=True)
