"""N training steps of a completion network (no timing, no profiler): the payload of `rocprofv3 --kernel-trace -- python tools/step_loop.py ecg 14`,
summarised by tools/ktrace_tail.py (steady-state steps only).  python tools/step_loop.py [vrcnet|ecg|pcn] [steps]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "completion"))
import torch, train
name = sys.argv[1] if len(sys.argv) > 1 else "ecg"
args = train.load_config(os.path.join(ROOT, "completion", "cfgs", name + ".yaml")); args.load_model = None
net = importlib.import_module("models." + name).Model(args).to("cuda:0").train()
opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)
g = torch.Generator().manual_seed(0)
gt = torch.rand(32, 2048, 3, generator=g).to("cuda:0"); partial = gt.transpose(2, 1).contiguous()
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    opt.zero_grad(); _, _, loss = net(partial, gt, alpha=0.5); loss.backward(); opt.step()
torch.cuda.synchronize()
