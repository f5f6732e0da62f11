"""Synthetic surface-shaped clouds for benchmarks and tests (no dataset can be downloaded here): MVP clouds are samples
of 2-manifolds (completion/dataset.py:21-34, completion/README.md:21-32), not uniform volumes.  Each generator takes a
seeded torch.Generator and returns (b, n, 3) float32 points inside [0, 1]^3 (the range the auction EMD expects,
utils/metrics/EMD/README.md:18)."""
import math

import torch


def sphere(g, b, n):
    v = torch.randn(b, n, 3, generator=g)
    return 0.5 + 0.4 * v / v.norm(dim=2, keepdim=True)


def torus(g, b, n):
    u, v = 2 * math.pi * torch.rand(b, n, generator=g), 2 * math.pi * torch.rand(b, n, generator=g)
    R, r = 0.3, 0.12
    return torch.stack([0.5 + (R + r * torch.cos(v)) * torch.cos(u), 0.5 + (R + r * torch.cos(v)) * torch.sin(u),
                        0.5 + r * torch.sin(v)], 2)


def box(g, b, n):
    p = torch.rand(b, n, 3, generator=g)
    face = torch.randint(0, 6, (b, n), generator=g)
    axis, side = face % 3, (face // 3).float()
    p.scatter_(2, axis.unsqueeze(2), side.unsqueeze(2))
    return 0.15 + 0.7 * p


def chair(g, b, n):
    """A crude MVP-like shape: seat + back (thin slabs' surfaces) + four legs (thin cylinders)."""
    part = torch.rand(b, n, generator=g)
    p = torch.rand(b, n, 3, generator=g)
    seat = torch.stack([0.2 + 0.6 * p[..., 0], 0.2 + 0.6 * p[..., 1], 0.45 + 0.04 * (p[..., 2] > 0.5).float()], 2)
    back = torch.stack([0.2 + 0.6 * p[..., 0], 0.76 + 0.04 * (p[..., 1] > 0.5).float(), 0.49 + 0.4 * p[..., 2]], 2)
    ang = 2 * math.pi * p[..., 0]
    leg = (p[..., 1] * 4).long().clamp(max=3)
    cx, cy = 0.25 + 0.5 * (leg % 2).float(), 0.25 + 0.5 * (leg // 2).float()
    legs = torch.stack([cx + 0.02 * torch.cos(ang), cy + 0.02 * torch.sin(ang), 0.05 + 0.4 * p[..., 2]], 2)
    return torch.where((part < 0.45).unsqueeze(2), seat, torch.where((part < 0.8).unsqueeze(2), back, legs))


SHAPES = {"sphere": sphere, "torus": torus, "box": box, "chair": chair}


def prediction_pair(name, mode, g, b, n):
    """(pred, gt) of shape `name`: mode "indep" = two independent samples of the surface; a float sigma = gt + Gaussian
    noise of that size (what a trained completion network's output looks like next to its target)."""
    gt = SHAPES[name](g, b, n)
    pred = SHAPES[name](g, b, n) if mode == "indep" else gt + float(mode) * torch.randn(b, n, 3, generator=g)
    return pred.float().contiguous(), gt.float().contiguous()
