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


def run_ex(cases, seed, verbose=True):
  """mvp_pointwise_mfma_ex / mvp_pointwise_wgrad_mfma_ex (ABI 18) on random ragged shapes: every fused step against the same
  GEMM between separate torch passes, BIT FOR BIT (ReLU of x on load, a bias per cloud, + residual then ReLU, the residual as
  a mask, two outputs at every multiple of 32, x as relu(x) in the weight gradient).  -> number of failing cases"""
  g = torch.Generator().manual_seed(seed)
  ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
  bad = 0
  for case in range(cases):
      B = ri(1, 7); L = 4 * ri(1, 200) if ri(0, 3) else 4 * ri(1, 4)
      cin = ri(1, 300) if ri(0, 2) else ri(1, 9); cout = ri(1, 400) if ri(0, 2) else ri(1, 9)
      x = torch.randn(B, cin, L, generator=g).to(dev); w = torch.randn(cout, cin, generator=g).to(dev)
      b = torch.randn(cout, generator=g).to(dev) if ri(0, 1) else None
      cb = torch.randn(B, cout, generator=g).to(dev)
      res = torch.randn(B, cout, L, generator=g).to(dev)
      bb = b.view(1, -1, 1) if b is not None else 0
      plain = mfma_linear(x, w)
      plain_r = mfma_linear(torch.relu(x), w)
      ok = {}
      ok["x_relu"] = torch.equal(mfma_linear(x, w, b, x_relu=True), plain_r + bb)
      ok["x_relu+relu"] = torch.equal(mfma_linear(x, w, b, x_relu=True, relu=True), torch.relu(plain_r + bb))
      ok["cloud_bias"] = torch.equal(mfma_linear(x, w, cb, bias_per_cloud=True, relu=bool(case & 1)),
                                     torch.relu(plain + cb.unsqueeze(2)) if case & 1 else plain + cb.unsqueeze(2))
      ok["res+relu"] = torch.equal(mfma_linear(x, w, b, residual=res, relu_after=True), torch.relu((plain + bb) + res))
      ok["x_relu+res+relu"] = torch.equal(mfma_linear(x, w, b, x_relu=True, residual=res, relu_after=True), torch.relu((plain_r + bb) + res))
      ok["mask"] = torch.equal(mfma_linear(x, w, b, residual=res, res_is_mask=True), torch.where(res > 0, plain + bb, torch.zeros_like(plain)))
      for split in range(32, cout, 32):
          if ri(0, 2) == 0 or split == 32:
              y1, y2 = mfma_linear(x, w, b, m_split=split)
              full = plain + bb
              ok["split%d" % split] = torch.equal(y1, full[:, :split]) and torch.equal(y2, full[:, split:]) and y1.is_contiguous() and y2.is_contiguous()
      if cin % 4 == 0:
          gy = torch.randn(B, cout, L, generator=g).to(dev)
          ok["dgrad_mask"] = torch.equal(mfma_linear(gy, w, w_kmajor=True, residual=x, res_is_mask=True),
                                         torch.where(x > 0, mfma_linear(gy, w, w_kmajor=True), torch.zeros_like(x)))
          ymask = torch.randn(B, cout, L, generator=g).to(dev)
          for m in (None, ymask):
              gw, gb = mfma_wgrad(x, gy, cout, cin, True, gymask=m, x_relu=True)
              gw2, gb2 = mfma_wgrad(torch.relu(x), gy, cout, cin, True, gymask=m)
              ok["wgrad_x_relu" + ("_masked" if m is not None else "")] = torch.equal(gw, gw2) and torch.equal(gb, gb2)
      fails = [k for k, v in ok.items() if not v]
      if fails:
          bad += 1
          print("FAIL", (B, cin, cout, L), "bias", b is not None, fails, flush=True)
  if verbose:
    print("ex: %d cases, %d failed (seed %d)" % (cases, bad, seed))
  return bad


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "ex":
        sys.exit(1 if run_ex(int(sys.argv[1]), int(sys.argv[2])) else 0)

    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
