"""Training step of ECG / VRCNet: eager vs captured into ONE HIP graph (torch.cuda.CUDAGraph), with the
1x1 convolutions on the library or routed to the MFMA kernels (pointwise.MFMA_TRAIN) -- the four cells
VERDICT r2 asked for.  forward + CD loss + backward + Adam (capturable) on static input buffers.
   python tools/bench_graph_step.py [ecg|vrcnet ...]        (MVP_BENCH_REPS: timed steps per cell)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "completion"))
import torch
import train
import mvp_benchmark_amd.pointwise as pw

REPS = int(os.environ.get("MVP_BENCH_REPS", "20"))
dev = "cuda:0"


def timed(fn, reps=REPS):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for name in sys.argv[1:] or ("vrcnet", "ecg"):
    for routed in (False, True):
        pw.MFMA_TRAIN = routed
        g = torch.Generator().manual_seed(0)
        torch.manual_seed(0)
        args = train.load_config(os.path.join(ROOT, "completion", "cfgs", name + ".yaml"))
        args.load_model = None
        net = importlib.import_module("models." + name).Model(args).to(dev).train()
        opt = torch.optim.Adam(net.parameters(), lr=1e-4, capturable=True, fused=True)
        gt = torch.rand(32, 2048, 3, generator=g).to(dev)
        partial = gt.transpose(2, 1).contiguous()

        def step():
            opt.zero_grad(set_to_none=True)
            _, _, loss = net(partial, gt, alpha=0.5)
            loss.backward()
            opt.step()
            return loss

        eager = timed(step)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph):
                static_loss = step()
            graph.replay(); torch.cuda.synchronize()
            l1 = float(static_loss)
            graphed = timed(graph.replay)
            print("%s train step (batch 32, 2048 pts), 1x1 convolutions %s: eager %.2f ms, one HIP graph %.2f ms "
                  "(loss after a replay %.6f)" % (name, "MFMA-routed" if routed else "library", eager, graphed, l1), flush=True)
        except Exception as e:  # noqa: BLE001
            print("%s (%s): eager %.2f ms; capture failed: %s: %s" % (
                name, "MFMA-routed" if routed else "library", eager, type(e).__name__, str(e).splitlines()[0][:300]), flush=True)
        del graph, net, opt
        torch.cuda.empty_cache()
