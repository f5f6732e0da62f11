"""Elementwise / copy / cat operators of one VRCNet / ECG training step by input shapes (torch profiler, record_shapes):
which tensors the ~5 ms of glue kernels move.  python tools/profile_aten_shapes.py [vrcnet|ecg]"""
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
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
        step()
        torch.cuda.synchronize()
    rows = [(e.key, str(e.input_shapes)[:110], e.self_device_time_total / 1e3, e.count) for e in prof.key_averages(group_by_input_shape=True)
            if e.self_device_time_total > 0 and e.key.startswith("aten::")]
    print("===== %s: aten operators by input shapes (ms per step, calls)" % name)
    for r in sorted(rows, key=lambda r: -r[2])[:45]:
        print("%-28s %-112s %7.3f ms x%d" % r)
