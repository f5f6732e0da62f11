"""Same-box, same-process A/B of op_config switches on a training step (the box-to-box spread of the
model benches is larger than the changes under test): ONE network and optimizer, the two settings alternated ROUNDS times,
REPS timed steps each after two untimed ones; per-round times, their medians and the number of launching operator calls.
python tools/ab_vrcnet.py "singleton_sk=0" "singleton_sk=1" [vrcnet|ecg] [rounds]"""
import importlib, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "completion"))
import torch
import train
import op_config

def parse(s):
    return {k: int(v) for k, v in (kv.split("=") for kv in s.split(",") if kv)}

A, B = parse(sys.argv[1]), parse(sys.argv[2])
name = sys.argv[3] if len(sys.argv) > 3 else "vrcnet"
ROUNDS = int(sys.argv[4]) if len(sys.argv) > 4 else 6
REPS = int(os.environ.get("MVP_BENCH_REPS", "10"))
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
args = train.load_config(os.path.join(ROOT, "completion", "cfgs", name + ".yaml")); args.load_model = None
net = importlib.import_module("models." + name).Model(args).to(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)
gt = torch.rand(32, 2048, 3, generator=g).to(dev); partial = gt.transpose(2, 1).contiguous()

def step():
    opt.zero_grad(); _, _, loss = net(partial, gt, alpha=0.5); loss.backward(); opt.step()

def timed(setting):
    op_config.OPS.reset(); op_config.configure(**setting)
    step(); step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / REPS * 1e3

def launches(setting):
    from torch.profiler import profile, ProfilerActivity
    op_config.OPS.reset(); op_config.configure(**setting)
    step(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step(); torch.cuda.synchronize()
    return sum(1 for e in prof.events() if str(e.device_type).endswith("CUDA"))

ta, tb = [], []
for r in range(ROUNDS):
    ta.append(timed(A)); tb.append(timed(B))
print("%s step, %d alternations x %d steps" % (name, ROUNDS, REPS))
print("A %-40s %s  median %.2f ms" % (A, " ".join("%.2f" % t for t in ta), statistics.median(ta)))
print("B %-40s %s  median %.2f ms" % (B, " ".join("%.2f" % t for t in tb), statistics.median(tb)))
print("B - A: %+.2f ms (%+.1f %%)" % (statistics.median(tb) - statistics.median(ta),
                                     100 * (statistics.median(tb) / statistics.median(ta) - 1)))
try:
    print("kernel launches per step: A %d, B %d" % (launches(A), launches(B)))
except Exception as e:      # (the profiler is optional)
    print("launch count unavailable:", e)
