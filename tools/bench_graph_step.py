"""Training step of ECG / VRCNet captured into one HIP graph (torch.cuda.CUDAGraph) against the eager step:
forward + CD loss + backward + Adam (capturable) on static input buffers.  python tools/bench_graph_step.py [ecg|vrcnet]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "completion"))
import torch
import train

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
for name in sys.argv[1:] or ("ecg", "vrcnet"):
    args = train.load_config(os.path.join(ROOT, "completion", "cfgs", name + ".yaml"))
    args.load_model = None
    net = importlib.import_module("models." + name).Model(args).to(dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, capturable=True)
    gt = torch.rand(32, 2048, 3, generator=g).to(dev)
    partial = gt.transpose(2, 1).contiguous()
    def step():
        opt.zero_grad(set_to_none=True)
        _, _, loss = net(partial, gt, alpha=0.5)
        loss.backward()
        opt.step()
        return loss
    def timed(fn, reps=5):
        fn(); fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
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
        print("%s train step (batch 32): eager %.1f ms, one HIP graph %.1f ms (loss after replay %.6f)" % (name, eager, graphed, l1), flush=True)
    except Exception as e:  # noqa: BLE001
        print("%s: capture failed: %s: %s" % (name, type(e).__name__, str(e).splitlines()[0][:300]), flush=True)
