"""Randomised check of the 1x1-convolution kernels (csrc/pointwise_mfma.hip, pointwise_max.hip) against float64 PyTorch:
random (B, Cin, Cout, L) incl. 1-channel layers, ragged tiles, L = 4, masks, bias on / off, padded weight rows.
python tools/fuzz_pointwise.py [cases] [seed]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvp_benchmark_amd.pointwise import mfma_linear, mfma_wgrad, PointwiseConv1d
dev = "cuda:0"


def run(cases, seed, verbose=True):
  """-> number of failing cases"""
  g = torch.Generator().manual_seed(seed)
  ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
  bad = 0
  for case in range(cases):
      B = ri(1, 9); L = 4 * ri(1, 300) if ri(0, 3) else 4 * ri(1, 4)
      cin = ri(1, 700) if ri(0, 2) else ri(1, 9); cout = ri(1, 600) if ri(0, 2) else ri(1, 9)
      x = torch.randn(B, cin, L, generator=g).to(dev); w = torch.randn(cout, cin, generator=g).to(dev)
      b = torch.randn(cout, generator=g).to(dev) if ri(0, 1) else None
      gy = torch.randn(B, cout, L, generator=g).to(dev); ymask = torch.randn(B, cout, L, generator=g).to(dev)
      relu = bool(ri(0, 1))
      xd, wd, gd = x.double(), w.double(), gy.double()
      ref = torch.einsum("oc,bcl->bol", wd, xd) + (b.double().view(1, -1, 1) if b is not None else 0)
      ref = torch.relu(ref) if relu else ref
      errs = {}
      errs["fwd"] = ((mfma_linear(x, w, b, relu=relu).double() - ref).abs().max().item(), 1e-5 * math.sqrt(cin) * 4)
      if cin % 4 == 0:
          gm = gd * (ymask.double() > 0)
          want = torch.einsum("oc,bol->bcl", wd, gm)
          errs["dgrad"] = ((mfma_linear(gy, w, w_kmajor=True, xmask=ymask).double() - want).abs().max().item(), 1e-5 * math.sqrt(cout) * 4)
      use_mask = bool(ri(0, 1))
      gm = gd * (ymask.double() > 0) if use_mask else gd
      gw, gb = mfma_wgrad(x, gy, cout, cin, b is not None, gymask=ymask if use_mask else None)
      tol = 1e-5 * math.sqrt(B * L) * 4
      errs["wgrad"] = ((gw.double() - torch.einsum("bol,bil->oi", gm, xd)).abs().max().item(), tol)
      if b is not None:
          errs["bgrad"] = ((gb.double() - gm.sum((0, 2))).abs().max().item(), tol)
      # conv -> max backward
      layer = PointwiseConv1d(cin, cout, bias=b is not None).to(dev)
      xr = x.clone().requires_grad_()
      go = torch.randn(B, cout, generator=g).to(dev)
      params = (xr,) + tuple(layer.parameters())
      got = torch.autograd.grad(layer.max_over_positions(xr), params, go)
      y = layer(xr); val, idx = y.max(dim=2)
      cols = torch.gather(xd, 2, idx.unsqueeze(1).expand(B, cin, cout))
      want_w = torch.einsum("bo,bio->oi", go.double(), cols)
      want_x = torch.zeros_like(xd).scatter_add_(2, idx.unsqueeze(1).expand(B, cin, cout), go.double().unsqueeze(1) * layer.weight.detach().double().view(cout, cin).t().unsqueeze(0))
      errs["max_gw"] = ((got[1].double().view(cout, cin) - want_w).abs().max().item(), 1e-5 * max(1.0, float(want_w.abs().max())))
      errs["max_gx"] = ((got[0].double() - want_x).abs().max().item(), 1e-5 * max(1.0, float(want_x.abs().max())))
      fails = {k: v for k, v in errs.items() if not (v[0] < v[1])}
      if fails:
          bad += 1
          print("FAIL", (B, cin, cout, L), "bias", b is not None, "relu", relu, fails, flush=True)
  if verbose:
    print("%d cases, %d failed (seed %d)" % (cases, bad, seed))
  return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
