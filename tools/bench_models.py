"""End-to-end model steps at the BASELINE configs (single GPU):
   cfg 2: PCN eval 2048 -> 16384 pts, batch 32 (CD + F1 + EMD)
   cfg 3: VRCNet train step, per-rank batch 32 (2048 pts), CD loss
   (+ ECG train step).  Synthetic data, random-init weights."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "completion"))
import torch
import train
import mvp_benchmark_amd.pointwise as pw
if "MVP_MFMA_TRAIN" in os.environ:                       # A/B of the training-path routing (pointwise.py)
    pw.MFMA_TRAIN = os.environ["MVP_MFMA_TRAIN"] == "1"
import op_config
# A/B of the networks' formulations: MVP_OPS="gather_sum=0,side_lanes=1" (completion/op_config.py)
OPS_OVERRIDE = {k: int(v) for k, v in (kv.split("=") for kv in os.environ["MVP_OPS"].split(","))} if os.environ.get("MVP_OPS") else {}


def load_cfg(path):
    """train.load_config sets the op-layer switches from the cfg (else the defaults); the A/B override goes on top."""
    args = train.load_config(path)
    if OPS_OVERRIDE:
        op_config.configure(**OPS_OVERRIDE)
    return args


REPS = int(os.environ.get("MVP_BENCH_REPS", "10"))
if os.environ.get("MVP_CUDNN_BENCHMARK") == "1":          # A/B: MIOpen picks its convolution solvers by measuring them
    torch.backends.cudnn.benchmark = True

dev = "cuda:0"
g = torch.Generator().manual_seed(0)

def timed(fn, reps=REPS):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

# cfg 2 (the randomly initialised PCN emits one tight blob: the degenerate EMD input, seconds per step;
# pass "cfg2" to include it)
if "cfg2" in sys.argv:
    args = load_cfg(os.path.join(ROOT, "completion", "cfgs", "pcn_eval16k.yaml"))
    net = importlib.import_module("models.pcn").Model(args).to(dev).eval()
    partial = torch.rand(32, 3, 2048, generator=g).to(dev); gt = torch.rand(32, 16384, 3, generator=g).to(dev)
    with torch.no_grad():
        ms = timed(lambda: net(partial, gt, prefix="val"), reps=1)
        args.eval_emd = False; net.eval_emd = False
        ms_noemd = timed(lambda: net(partial, gt, prefix="val"))
    print("cfg2 PCN eval (32, 2048->16384) CD+F1+EMD: %.1f ms/step (%.1f clouds/s); without EMD %.1f ms" % (ms, 32e3 / ms, ms_noemd), flush=True)

for name in ("vrcnet", "ecg", "pcn"):
    args = load_cfg(os.path.join(ROOT, "completion", "cfgs", name + ".yaml"))
    args.load_model = None
    net = importlib.import_module("models." + name).Model(args).to(dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=os.environ.get("MVP_FOREACH_ADAM") != "1")   # (fused: as completion/train.py builds it)
    gt = torch.rand(32, 2048, 3, generator=g).to(dev)
    partial = gt.transpose(2, 1).contiguous()
    def step():
        opt.zero_grad()
        _, _, loss = net(partial, gt, alpha=0.5)
        loss.backward()
        opt.step()
    if torch.backends.cudnn.benchmark:
        for _ in range(3): step()       # the solver search happens in the first steps
    # (ECG settles late: its first ~150 steps run 20-25 ms while the libraries pick their kernels -- tools/ab_vrcnet.py ecg 14
    # shows the same curve in both settings, profiles/r6b_vrcnet_ab.txt -- so every model gets the same long warm-up)
    for _ in range(int(os.environ.get("MVP_BENCH_WARMUP", "150"))):
        step()
    ms = timed(step)
    print("%s train step (batch 32, 2048 pts, MFMA_TRAIN=%s%s): %.1f ms/step (%.1f samples/s); grads %.1f MB fp32" % (
        name, pw.MFMA_TRAIN, ", solver search on" if torch.backends.cudnn.benchmark else "", ms, 32e3 / ms, sum(p.numel() for p in net.parameters()) * 4 / 1e6), flush=True)
