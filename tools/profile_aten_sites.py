"""Where the elementwise / copy launches of a training step come from: aten operators of one VRCNet / ECG step grouped by
(operator, the innermost Python frame inside completion/ or mvp_benchmark_amd/), by the GPU time of their own kernels.
Backward operators have no Python frame (autograd thread): they are grouped by name alone.
python tools/profile_aten_sites.py [vrcnet|ecg]"""
import collections, importlib, os, sys
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
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0.0, 0])
    for e in prof.events():
        if str(e.device_type).endswith("CUDA") or e.self_device_time_total <= 0:
            continue
        site = "-"
        for fr in e.stack or []:
            if ("/completion/" in fr or "/mvp_benchmark_amd/" in fr) and "site-packages" not in fr:
                site = fr.replace(ROOT + "/", "")
                break
        a = agg[(e.name, site)]
        a[0] += e.self_device_time_total / 1e3
        a[1] += 1
    rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
    print("===== %s: %.2f ms, %d launching operator calls" % (name, sum(v[0] for _, v in rows), sum(v[1] for _, v in rows)))
    for (op, site), (ms, cnt) in rows[:90]:
        print("%7.3f ms x%-3d %-42s %s" % (ms, cnt, op[:42], site[:110]))
