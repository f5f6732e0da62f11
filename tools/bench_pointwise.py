"""Per-point / per-edge linear maps of the completion networks: MIOpen 1x1 convolution against a batched GEMM on
the same (B, C, L) layout.  Collects the shapes from one VRCNet (or ECG) step with forward hooks, then times
forward + backward of every distinct shape both ways.   python tools/bench_pointwise.py [vrcnet|ecg]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "completion"))
import torch
import torch.nn as nn
import torch.nn.functional as F
import train

dev = "cuda:0"
name = sys.argv[1] if len(sys.argv) > 1 else "vrcnet"
g = torch.Generator().manual_seed(0)
args = train.load_config(os.path.join(ROOT, "completion", "cfgs", name + ".yaml"))
args.load_model = None
net = importlib.import_module("models." + name).Model(args).to(dev).train()
shapes = {}
def hook(m, inp, out):
    x = inp[0]
    key = (type(m).__name__, m.in_channels, m.out_channels, tuple(x.shape), m.bias is not None, x.is_contiguous())
    shapes[key] = shapes.get(key, 0) + 1
for m in net.modules():
    if isinstance(m, (nn.Conv1d, nn.Conv2d)):
        m.register_forward_hook(hook)
gt = torch.rand(32, 2048, 3, generator=g).to(dev)
partial = gt.transpose(2, 1).contiguous()
_, _, loss = net(partial, gt, alpha=0.5)
loss.backward()
torch.cuda.synchronize()

def timed(fn, reps=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

tot_c = tot_g = 0.0
for key, cnt in sorted(shapes.items(), key=lambda kv: -kv[0][1] * kv[0][2] * kv[1]):
    kind, cin, cout, shp, bias, contig = key
    x = torch.randn(*shp, device=dev, requires_grad=True)
    w = torch.randn(cout, cin, *([1] * (len(shp) - 2)), device=dev, requires_grad=True)
    b = torch.randn(cout, device=dev, requires_grad=True) if bias else None
    conv = F.conv1d if len(shp) == 3 else F.conv2d
    def f_conv():
        y = conv(x, w, b)
        y.backward(y)
    def f_gemm():
        y = torch.matmul(w.flatten(1), x.flatten(2))
        if b is not None:
            y = y + b[:, None]
        y = y.view(shp[0], cout, *shp[2:])
        y.backward(y)
    tc, tg = timed(f_conv), timed(f_gemm)
    t1 = None
    if len(shp) == 4 and shp[2] == 1:   # (B, C, 1, N) through the 1-d convolution
        def f_conv1d():
            y = F.conv1d(x.squeeze(2), w.squeeze(3), b).unsqueeze(2)
            y.backward(y)
        t1 = timed(f_conv1d)
    L = 1
    for d in shp[2:]:
        L *= d
    gf = 3 * 2.0 * shp[0] * L * cin * cout / 1e9
    tot_c += tc * cnt; tot_g += tg * cnt
    print("%s %4d->%4d x%-2d in %-22s contig %d: conv %.3f ms  gemm %.3f ms  (%.1f GFLOP fwd+bwd, gemm %.1f TF/s)%s" % (
        kind, cin, cout, cnt, shp, contig, tc, tg, gf, gf / tg, "" if t1 is None else "  as conv1d %.3f ms" % t1), flush=True)
    if t1 is not None:
        tot_1 = globals().get("tot_1", 0.0) + (t1 - tc) * cnt
        globals()["tot_1"] = tot_1
print("%s: all pointwise layers fwd+bwd: conv %.2f ms, gemm %.2f ms" % (name, tot_c, tot_g))
print("conv2d (B,C,1,N) layers routed through conv1d: %+.2f ms in total" % globals().get("tot_1", 0.0))
