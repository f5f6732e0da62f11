"""Every aten operator call of ONE training step with the Python site that issued it (innermost frame inside completion/ or
mvp_benchmark_amd/), the bytes of its tensor arguments and results, forward and backward (backward calls carry the site
"<backward>" + the autograd node's name) -- the launch census the torch profiler's with_stack does not give on this build.
python tools/dispatch_sites.py [vrcnet|ecg]"""
import collections, importlib, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "completion"))
import torch
import train
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils._pytree import tree_flatten

dev = "cuda:0"
name = sys.argv[1] if len(sys.argv) > 1 else "vrcnet"
g = torch.Generator().manual_seed(0)
args = train.load_config(os.path.join(ROOT, "completion", "cfgs", name + ".yaml")); args.load_model = None
net = importlib.import_module("models." + name).Model(args).to(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)
gt = torch.rand(32, 2048, 3, generator=g).to(dev); partial = gt.transpose(2, 1).contiguous()
SKIP = ("aten.view", "aten.t.", "aten.transpose", "aten.unsqueeze", "aten.squeeze", "aten.expand", "aten.detach", "aten.alias",
        "aten.slice", "aten.select", "aten.split", "aten._unsafe_view", "aten.permute", "aten.as_strided", "aten.empty", "aten.reshape",
        "aten.unbind", "aten.chunk", "aten.is_", "aten.sym_", "aten.lift", "aten._reshape_alias", "aten.narrow", "aten.stride", "aten.size")
rows = collections.defaultdict(lambda: [0, 0])

class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, a=(), kw=None):
        out = func(*a, **(kw or {}))
        nm = str(func)
        if nm.startswith(SKIP):
            return out
        site = None
        for fr in reversed(traceback.extract_stack()):
            fn = fr.filename
            if ("/completion/" in fn or "/mvp_benchmark_amd/" in fn) and "/tools/" not in fn:
                site = "%s:%d %s" % (fn.split("/completion/")[-1].split("/mvp_benchmark_amd/")[-1], fr.lineno, fr.name)
                break
        if site is None:
            site = "<backward / optimizer>"
        nbytes = sum(t.numel() * t.element_size() for t in tree_flatten((a, kw, out))[0] if isinstance(t, torch.Tensor))
        r = rows[(nm, site)]
        r[0] += 1; r[1] += nbytes
        return out

def step():
    opt.zero_grad(); _, _, loss = net(partial, gt, alpha=0.5); loss.backward(); opt.step()
step(); step(); torch.cuda.synchronize()
with Census():
    step()
torch.cuda.synchronize()
tot = sum(v[0] for v in rows.values())
print("===== %s: %d dispatched non-view operator calls in one step; by bytes touched" % (name, tot))
for (nm, site), (cnt, nbytes) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:140]:
    print("%9.1f MB x%-3d %-38s %s" % (nbytes / 1e6, cnt, nm[:38], site))
