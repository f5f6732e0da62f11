"""GPU report (not collected by pytest): a training step of this repo's PCN / VRCNet / ECG code (a) as shipped, (b) in the
REFERENCE'S FORMULATION (completion/op_config.py: every rewrite off, PyTorch's convolutions instead of the MFMA layer, foreach
Adam) on this repo's operator kernels, (c) the same on the REFERENCE'S OWN operator kernels (oracle/_ref via tests/ref_ops.py).
(c) is the closest this box gets to "the reference's step on an MI355X": the model files are this repo's (same parameters, same
mathematics: tests/test_model_golden.py), the formulation and the kernels under it are the reference's.  ECG's edge
convolutions keep this repo's split form in (b) / (c) (no switch): its (c) is a lower bound of the reference's cost.

    python tests/report_reference_model_step.py [pcn vrcnet ecg]

Lives under tests/ because it executes oracle/ code (test infrastructure only)."""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "completion"))

import train  # noqa: E402
import op_config  # noqa: E402
import mvp_benchmark_amd.pointwise as pw  # noqa: E402
import ref_ops  # noqa: E402

dev = "cuda:0"


def modules():
    import model_utils
    from models import _common, ecg, edge_unet, pcn, relational, vrcnet
    return [model_utils, _common, relational, edge_unet, ecg, vrcnet, pcn]


def timed(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def step_ms(name, fused, switches=None):
    g = torch.Generator().manual_seed(0)
    args = train.load_config(os.path.join(ROOT, "completion", "cfgs", name + ".yaml"))
    args.load_model = None
    if switches:                         # (load_config applies the cfg's switches, else the defaults: override afterwards)
        op_config.configure(**switches)
    torch.manual_seed(0)                 # the same initial weights in every mode
    net = importlib.import_module("models." + name).Model(args).to(dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=fused)
    gt = torch.rand(32, 2048, 3, generator=g).to(dev)
    partial = gt.transpose(2, 1).contiguous()

    losses = []

    def step():
        opt.zero_grad()
        torch.manual_seed(1)             # (VRCNet's reparameterisation noise / dropout: the same draw in every mode)
        _, _, loss = net(partial, gt, alpha=0.5)
        if len(losses) < 1:
            losses.append(float(loss.mean()))
        loss.backward()
        opt.step()
    return timed(step), losses[0]


def main():
    names = sys.argv[1:] or ["pcn", "vrcnet", "ecg"]
    print("train step, batch 32 x 2048 points, ms (%s)" % torch.cuda.get_device_name(0))
    for name in names:
        op_config.OPS.reset()
        pw.MFMA_TRAIN = True
        a, la = step_ms(name, True)
        off = dict(gather_sum=0, gather_max=0, side_lanes=0, stacked_projections=0, skip_full_fps_of_gt=0,
                   conv_before_interp=0, folded_conv=0)
        pw.MFMA_TRAIN = False
        b, lb = step_ms(name, False, off)
        undo = ref_ops.patch_ops(modules())
        ref_ops.ref.SYNC = False            # like the reference's wrappers: no host synchronisation per operator
        try:
            c, lc = step_ms(name, False, off)
        finally:
            ref_ops.ref.SYNC = True
            undo()
        print("  %-7s as shipped %7.2f | reference formulation, this repo's op kernels %7.2f | reference formulation, the reference's "
              "op kernels %7.2f   (%.1fx)   first-step loss %.7f | %.7f | %.7f" % (name, a, b, c, c / a, la, lb, lc), flush=True)
    op_config.OPS.reset()
    pw.MFMA_TRAIN = True


if __name__ == "__main__":
    main()
