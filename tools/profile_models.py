"""Where a VRCNet / ECG training step spends its GPU time (torch profiler, top kernels).
python tools/profile_models.py [vrcnet|ecg ...] [hires]   (hires: batch 8, 8192 output points)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "completion"))
import torch
import train
from torch.profiler import profile, ProfilerActivity

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
hires = "hires" in sys.argv
names = [a for a in sys.argv[1:] if a != "hires"]
for name in names or ("vrcnet", "ecg"):
    args = train.load_config(os.path.join(ROOT, "completion", "cfgs", name + ".yaml"))
    args.load_model = None
    if hires:
        args.num_points = 8192
    net = importlib.import_module("models." + name).Model(args).to(dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)
    gt = torch.rand(32, 2048, 3, generator=g).to(dev)
    partial = gt.transpose(2, 1).contiguous()
    if hires:
        gt = torch.rand(8, 8192, 3, generator=g).to(dev)
        partial = torch.rand(8, 3, 2048, generator=g).to(dev)
    def step():
        opt.zero_grad()
        _, _, loss = net(partial, gt, alpha=0.5)
        loss.backward()
        opt.step()
    step(); step(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step(); torch.cuda.synchronize()
    print("=====", name)
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=70))
