"""GPU kernels of one VRCNet / ECG training step, by total time (kernel rows of the torch profiler only).
python tools/profile_kernels.py [vrcnet|ecg]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "completion"))
import torch
import train
from torch.profiler import profile, ProfilerActivity
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
for name in sys.argv[1:] or ("vrcnet",):
    args = train.load_config(os.path.join(ROOT, "completion", "cfgs", name + ".yaml")); args.load_model = None
    net = importlib.import_module("models." + name).Model(args).to(dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)
    gt = torch.rand(32, 2048, 3, generator=g).to(dev); partial = gt.transpose(2, 1).contiguous()
    def step():
        opt.zero_grad(); _, _, loss = net(partial, gt, alpha=0.5); loss.backward(); opt.step()
    step(); step(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    rows = [(e.key, e.self_device_time_total / 3e3, e.count // 3) for e in prof.key_averages()
            if e.self_device_time_total > 0 and str(e.device_type).endswith("CUDA")]
    total = sum(r[1] for r in rows)
    print("===== %s: %.2f ms of kernels per step" % (name, total))
    for r in sorted(rows, key=lambda r: -r[1])[:70]:
        print("%-110s %7.3f ms x%d" % (r[0][:110], r[1], r[2]))
