"""1x1-convolution shapes of one VRCNet / ECG training step, and forward+backward time of each under three library
formulations: F.conv (MIOpen), torch.matmul (W @ x), explicit bmm with a batch-summed weight gradient.
python tools/bench_conv1x1_forms.py [vrcnet|ecg]"""
import importlib, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "completion"))
import torch, torch.nn.functional as F
import train
dev = "cuda:0"
name = sys.argv[1] if len(sys.argv) > 1 else "vrcnet"
args = train.load_config(os.path.join(ROOT, "completion", "cfgs", name + ".yaml")); args.load_model = None
net = importlib.import_module("models." + name).Model(args).to(dev).train()
g = torch.Generator().manual_seed(0)
gt = torch.rand(32, 2048, 3, generator=g).to(dev); partial = gt.transpose(2, 1).contiguous()
shapes = collections.Counter()
o1, o2 = F.conv1d, F.conv2d
def rec(orig):
    def f(x, w, b=None, *a, **k):
        if w.shape[2:].numel() == 1:
            shapes[(tuple(x.shape), w.shape[0], b is not None, x.requires_grad)] += 1
        return orig(x, w, b, *a, **k)
    return f
F.conv1d, F.conv2d = rec(o1), rec(o2)
torch.conv1d_orig = torch.conv1d
_, _, loss = net(partial, gt, alpha=0.5); loss.backward()
F.conv1d, F.conv2d = o1, o2
torch.cuda.synchronize()

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

class BmmConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w); ctx.hb = b is not None
        y = torch.matmul(w, x.flatten(2))
        if b is not None: y += b.view(1, -1, 1)
        return y.view(x.shape[0], w.shape[0], *x.shape[2:])
    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy3, x3 = gy.flatten(2), x.flatten(2)
        gx = torch.matmul(w.t(), gy3).view_as(x) if ctx.needs_input_grad[0] else None
        gw = torch.bmm(gy3, x3.transpose(1, 2)).sum(0)
        gb = gy3.sum((0, 2)) if ctx.hb else None
        return gx, gw, gb

class OneGemmConv(torch.autograd.Function):
    """weight gradient as ONE GEMM over K = B*L: gy (B,Co,L) -> (Co, B*L) needs a transpose copy; measure it."""
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w); ctx.hb = b is not None
        y = torch.matmul(w, x.flatten(2))
        if b is not None: y += b.view(1, -1, 1)
        return y.view(x.shape[0], w.shape[0], *x.shape[2:])
    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy3, x3 = gy.flatten(2), x.flatten(2)
        gx = torch.matmul(w.t(), gy3).view_as(x) if ctx.needs_input_grad[0] else None
        gw = torch.einsum("bol,bil->oi", gy3, x3)
        gb = gy3.sum((0, 2)) if ctx.hb else None
        return gx, gw, gb

tot = collections.Counter()
print("%-28s %5s %3s %5s | %8s %8s %8s %8s" % ("x", "cout", "n", "xgrad", "conv", "matmul", "bmm-fn", "einsum"))
for (xs, co, hb, xg), cnt in sorted(shapes.items(), key=lambda kv: -kv[0][0][0] * kv[0][0][1] * kv[1]):
    ci = xs[1]
    x = torch.randn(*xs, device=dev, requires_grad=xg)
    w = torch.randn(co, ci, device=dev, requires_grad=True)
    b = torch.randn(co, device=dev, requires_grad=True) if hb else None
    wc = w.view(co, ci, *([1] * (len(xs) - 2)))
    conv = o1 if len(xs) == 3 else o2
    gy = torch.randn(xs[0], co, *xs[2:], device=dev)
    def run(f):
        def go():
            y = f(); y.backward(gy)
            w.grad = None
            if xg: x.grad = None
        return go
    t_conv = timeit(run(lambda: conv(x, wc, b)))
    def mm():
        y = torch.matmul(w, x.flatten(2))
        if hb: y = y + b.view(1, -1, 1)
        return y.view(xs[0], co, *xs[2:])
    t_mm = timeit(run(mm))
    t_bmm = timeit(run(lambda: BmmConv.apply(x, w, b)))
    t_es = timeit(run(lambda: OneGemmConv.apply(x, w, b)))
    for k, v in (("conv", t_conv), ("matmul", t_mm), ("bmm", t_bmm), ("einsum", t_es), ("best", min(t_conv, t_mm, t_bmm, t_es))):
        tot[k] += v * cnt
    print("%-28s %5d %3d %5s | %8.3f %8.3f %8.3f %8.3f" % (str(xs), co, cnt, xg, t_conv, t_mm, t_bmm, t_es))
print("per step (ms):", dict((k, round(v, 2)) for k, v in tot.items()))
