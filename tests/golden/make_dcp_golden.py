"""Generates tests/golden/dcp_golden.npz from the REFERENCE DCP model, imported
unmodified from /root/reference/registration (runs in the build container only;
the reference never travels to the GPU box).

Stored: parameter names + shapes of registration/models/dcp.py:Model, a small
input (src, tgt (2,64,3), T_gt), and the reference's outputs in eval mode on
CPU: T_12 and (loss, r_err, t_err, rmse, rt_mse).  Parameters are NOT stored:
both sides fill them with `fill_parameters` below (closed-form, seeded by the
parameter's position in the sorted state_dict), so the fixture stays small.

Shims needed to import / run the reference on a CPU-only box without h5py
(none of them changes the arithmetic): a stub `h5py` module; `torch.arange`
inside dcp.py ignores its device='cuda' argument (dcp.py:49-51);
`Tensor.cuda()` is the identity (train_utils.py:98-99).
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/registration"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dcp_golden.npz")


def fill_parameters(model, seed=7):
    """Deterministic parameters independent of construction order."""
    sd = model.state_dict()
    for i, name in enumerate(sorted(sd)):
        t = sd[name]
        g = torch.Generator().manual_seed(seed * 1000 + i)
        if name.endswith("num_batches_tracked"):
            continue
        if name.endswith("running_var"):
            t.copy_(0.5 + torch.rand(t.shape, generator=g))
        elif name.endswith("running_mean"):
            t.copy_(0.1 * torch.randn(t.shape, generator=g))
        elif name.endswith("reflect"):
            continue
        elif name.endswith("a_2") or (name.endswith("weight") and t.dim() == 1):
            t.copy_(1.0 + 0.1 * torch.randn(t.shape, generator=g))
        elif t.dim() == 1:
            t.copy_(0.05 * torch.randn(t.shape, generator=g))
        else:
            fan_in = t[0].numel()
            t.copy_(torch.randn(t.shape, generator=g) / fan_in ** 0.5)
    model.load_state_dict(sd)


def make_inputs():
    g = torch.Generator().manual_seed(3)
    src = torch.rand(2, 64, 3, generator=g) - 0.5
    ang = torch.tensor([0.4, -0.9])
    c, s = torch.cos(ang), torch.sin(ang)
    zero, one = torch.zeros(2), torch.ones(2)
    R = torch.stack([c, -s, zero, s, c, zero, zero, zero, one], dim=1).view(2, 3, 3)
    t = torch.tensor([[0.1, -0.2, 0.05], [-0.15, 0.1, 0.2]])
    tgt = src @ R.transpose(1, 2) + t.unsqueeze(1)
    T = torch.eye(4).repeat(2, 1, 1)
    T[:, :3, :3] = R
    T[:, :3, 3] = t
    return src, tgt, T


def main():
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "models"))
    import dcp  # the reference, unmodified

    class _TorchOnCpu:
        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def arange(*a, **k):
            k.pop("device", None)
            return torch.arange(*a, **k)

    dcp.torch = _TorchOnCpu()
    torch.Tensor.cuda = lambda self, *a, **k: self

    net = dcp.Model(types.SimpleNamespace())
    fill_parameters(net)
    net.eval()
    src, tgt, T_gt = make_inputs()
    with torch.no_grad():
        T_12 = net(src, tgt)
        loss, r_err, t_err, rmse, rt_mse = net(src, tgt, T_gt)
    names = sorted(net.state_dict())
    shapes = [list(net.state_dict()[n].shape) for n in names]
    np.savez_compressed(OUT, names=np.array(names), shapes=np.array([str(s) for s in shapes]),
                        src=src.numpy(), tgt=tgt.numpy(), T_gt=T_gt.numpy(), T_12=T_12.numpy(),
                        loss=loss.numpy(), r_err=r_err.numpy(), t_err=t_err.numpy(), rmse=rmse.numpy(),
                        rt_mse=rt_mse.numpy())
    print("wrote", OUT, "params", sum(p.numel() for p in net.parameters()), "T_12[0]", T_12[0])
    print("r_err", r_err, "t_err", t_err)


if __name__ == "__main__":
    main()
