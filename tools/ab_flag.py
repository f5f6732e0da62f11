"""Same-process A/B of a module-level Python flag on a training step (e.g. mvp_benchmark_amd.mm3d_pn2.functional:ShareGatherSum.ONE_LAUNCH_BACKWARD):
python tools/ab_flag.py <module>:<attr path> <A value> <B value> [vrcnet|ecg] [rounds]"""
import importlib, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "completion"))
import torch
import train
mod, path = sys.argv[1].split(":")
obj = importlib.import_module(mod)
*parents, leaf = path.split(".")
for a in parents:
    obj = getattr(obj, a)
A, B = int(sys.argv[2]), int(sys.argv[3])
name = sys.argv[4] if len(sys.argv) > 4 else "vrcnet"
ROUNDS = int(sys.argv[5]) if len(sys.argv) > 5 else 5
REPS = int(os.environ.get("MVP_BENCH_REPS", "20"))
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
args = train.load_config(os.path.join(ROOT, "completion", "cfgs", name + ".yaml")); args.load_model = None
net = importlib.import_module("models." + name).Model(args).to(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)
gt = torch.rand(32, 2048, 3, generator=g).to(dev); partial = gt.transpose(2, 1).contiguous()
def step():
    opt.zero_grad(); _, _, loss = net(partial, gt, alpha=0.5); loss.backward(); opt.step()
def timed(v):
    setattr(obj, leaf, type(getattr(obj, leaf))(v))
    step(); step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / REPS * 1e3
ta, tb = [], []
for r in range(ROUNDS):
    ta.append(timed(A)); tb.append(timed(B))
print("%s step, %s, %d alternations x %d steps" % (name, sys.argv[1], ROUNDS, REPS))
print("A = %d: %s  median %.2f ms" % (A, " ".join("%.2f" % t for t in ta), statistics.median(ta)))
print("B = %d: %s  median %.2f ms" % (B, " ".join("%.2f" % t for t in tb), statistics.median(tb)))
print("B - A: %+.2f ms" % (statistics.median(tb) - statistics.median(ta)))
