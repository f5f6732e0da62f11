"""The three passes (forward, data gradient, weight + bias gradient) of every >= 32-channel 1x1-convolution shape of one
VRCNet / ECG training step (shapes and counts: tools/bench_conv1x1_forms.py, profiles/r3_conv1x1_forms.txt), library
(MIOpen through aten) against csrc/pointwise_mfma.hip, one row per shape and the count-weighted sums per step.
    python tools/bench_conv_passes.py [vrcnet|ecg]            MVP_LIB=<other libmvpops.so> for an A/B of two builds"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from mvp_benchmark_amd import _lib
if os.environ.get('MVP_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['MVP_LIB'])
from mvp_benchmark_amd.pointwise import mfma_linear, mfma_wgrad
dev = "cuda:0"

name = sys.argv[1] if len(sys.argv) > 1 else "vrcnet"
MINCH = int(sys.argv[2]) if len(sys.argv) > 2 else 32      # 1: also the layers with fewer than 32 input or output channels


def record_shapes(name):
    """(B, Cin, L, Cout) -> count of the 1x1 convolutions of one training step (batch 32, 2048 points)."""
    import collections, importlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "completion"))
    import train
    args = train.load_config(os.path.join(root, "completion", "cfgs", name + ".yaml")); args.load_model = None
    net = importlib.import_module("models." + name).Model(args).to(dev).train()
    g = torch.Generator().manual_seed(0)
    gt = torch.rand(32, 2048, 3, generator=g).to(dev); partial = gt.transpose(2, 1).contiguous()
    shapes = collections.Counter()
    o1, o2 = F.conv1d, F.conv2d
    def rec(orig):
        def f(x, w, b=None, *a, **k):
            if w.shape[2:].numel() == 1:
                shapes[(x.shape[0], x.shape[1], x[0, 0].numel(), w.shape[0])] += 1
            return orig(x, w, b, *a, **k)
        return f
    F.conv1d, F.conv2d = rec(o1), rec(o2)
    try:
        out = net(partial, gt, alpha=0.5)
    finally:
        F.conv1d, F.conv2d = o1, o2
    torch.cuda.synchronize()
    return sorted(((k + (n,)) for k, n in shapes.items() if min(k[1], k[3]) >= MINCH and (MINCH >= 32 or min(k[1], k[3]) < 32) and k[2] % 4 == 0),
                  key=lambda s: -s[0] * s[1] * s[2] * s[3] * s[4])


from mvp_benchmark_amd import pointwise as _pw
_pw.MFMA_TRAIN = _pw.USE_MFMA = False      # record through the library route (F.conv*)
shapes = record_shapes(name)
_pw.USE_MFMA = True


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print("# %s: 1x1-convolution passes, library | mvp (ms; TFLOP/s of the mvp kernel); lib = %s" % (name, _lib.LIB_PATH))
print("%-26s %2s | %-23s | %-23s | %-23s" % ("(B,Cin->Cout,L)", "n", "fwd+bias+relu lib mvp TF", "dgrad(relu') lib mvp TF", "wgrad+bias lib mvp TF"))
tot = {k: 0.0 for k in ("fl", "fm", "dl", "dm", "wl", "wm")}
for (B, cin, L, cout, n) in shapes:
    x = torch.randn(B, cin, L, device=dev); w = torch.randn(cout, cin, device=dev); b = torch.randn(cout, device=dev)
    w3 = w.unsqueeze(2).contiguous(); gy = torch.randn(B, cout, L, device=dev)
    y = torch.relu(F.conv1d(x, w3, b))
    fl = 2.0 * B * cin * cout * L
    gym = gy * (y > 0)        # the library route masks once for both backward passes: timed once, counted in `dl`
    cb = lambda m: torch.ops.aten.convolution_backward(gym, x, w3, [cout], [1], [0], [1], False, [0], 1, m)
    t = {}
    tmask = timeit(lambda: gy * (y > 0))
    t["fl"] = timeit(lambda: torch.relu_(F.conv1d(x, w3, b)))
    t["fm"] = timeit(lambda: mfma_linear(x, w, b, relu=True))
    t["dl"] = timeit(lambda: cb([True, False, False])) + tmask
    t["dm"] = timeit(lambda: mfma_linear(gy, w, w_kmajor=True, xmask=y)) if cin % 4 == 0 else t["dl"]
    t["wl"] = timeit(lambda: cb([False, True, True]))
    t["wm"] = timeit(lambda: mfma_wgrad(x, gy, cout, cin, True, gymask=y))
    for k in tot:
        tot[k] += n * t[k]
    print("%-26s %2d | %6.3f %6.3f %6.1f    | %6.3f %6.3f %6.1f    | %6.3f %6.3f %6.1f" % (
        "(%d,%d->%d,%d)" % (B, cin, cout, L), n, t["fl"], t["fm"], fl / t["fm"] / 1e9, t["dl"], t["dm"], fl / t["dm"] / 1e9,
        t["wl"], t["wm"], fl / t["wm"] / 1e9), flush=True)
print("per step (ms): forward lib %.2f mvp %.2f | dgrad lib %.2f mvp %.2f | wgrad lib %.2f mvp %.2f | all lib %.2f mvp %.2f" % (
    tot["fl"], tot["fm"], tot["dl"], tot["dm"], tot["wl"], tot["wm"], tot["fl"] + tot["dl"] + tot["wl"],
    tot["fm"] + tot["dm"] + tot["wm"]))
