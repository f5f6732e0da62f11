"""ECG / VRCNet train step at the 8192-point output setting (scale 4: the EF_expansion heads are active).
Batch 8 per GPU, synthetic data, random-init weights."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "completion"))
import torch
import train

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
for name in ("ecg", "vrcnet"):
    args = train.load_config(os.path.join(ROOT, "completion", "cfgs", name + ".yaml"))
    args.load_model = None
    args.num_points = 8192
    net = importlib.import_module("models." + name).Model(args).to(dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    gt = torch.rand(8, 8192, 3, generator=g).to(dev)
    partial = torch.rand(8, 3, 2048, generator=g).to(dev)
    def step():
        opt.zero_grad()
        _, _, loss = net(partial, gt, alpha=0.5)
        loss.backward()
        opt.step()
    step(); step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print("%s train step (batch 8, 2048 -> 8192 pts): %.1f ms/step" % (name, ms), flush=True)
