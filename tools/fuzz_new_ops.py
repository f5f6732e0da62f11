"""Randomised shapes for the round-1j/k kernels against PyTorch formulations (scatter gradients via scatter_add,
Gram top-k via torch.topk, shared-weight aggregation via repeat/product/sum, chamfer backward via autograd on the
gathered distances).  python tools/fuzz_new_ops.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mvp_benchmark_amd.mm3d_pn2 import gather_points, grouping_operation, three_interpolate
from mvp_benchmark_amd.mm3d_pn2.functional import gram_topk, share_weighted_sum
from mvp_benchmark_amd.metrics import cd

dev = "cuda:0"
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
g = torch.Generator().manual_seed(1234)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
t0, cases = time.time(), 0
while time.time() - t0 < budget:
    kind = ri(0, 4)
    if kind == 0:      # gather / group gradient
        b, c, n, p, s = ri(1, 4), ri(1, 40), ri(1, 9000), ri(1, 3000), ri(1, 20)
        hub = ri(0, 3) == 0
        f = torch.randn(b, c, n, generator=g).to(dev).requires_grad_()
        idx = torch.randint(0, min(n, 3) if hub else n, (b, p, s), generator=g).int().to(dev)
        go = torch.randn(b, c, p, s, generator=g).to(dev)
        got, = torch.autograd.grad(grouping_operation(f, idx), f, go)
        want = torch.zeros(b, c, n, device=dev)
        want.scatter_add_(2, idx.reshape(b, 1, -1).long().expand(-1, c, -1), go.reshape(b, c, -1))
        assert torch.allclose(got, want, rtol=1e-4, atol=1e-3 * (1 + p * s / max(1, min(n, 3) if hub else n)) ** 0.5), ("group grad", b, c, n, p, s, hub)
    elif kind == 1:    # three_interpolate gradient
        b, c, m, n = ri(1, 4), ri(1, 40), ri(1, 9000), ri(1, 4000)
        f = torch.randn(b, c, m, generator=g).to(dev).requires_grad_()
        idx = torch.randint(0, m, (b, n, 3), generator=g).int().to(dev)
        w = torch.rand(b, n, 3, generator=g).to(dev)
        go = torch.randn(b, c, n, generator=g).to(dev)
        got, = torch.autograd.grad(three_interpolate(f, idx, w), f, go)
        want = torch.zeros(b, c, m, device=dev)
        for r in range(3):
            want.scatter_add_(2, idx[:, :, r].reshape(b, 1, n).long().expand(-1, c, -1), go * w[:, :, r].unsqueeze(1))
        assert torch.allclose(got, want, rtol=1e-4, atol=1e-3 * (1 + 3 * n / m) ** 0.5), ("interp grad", b, c, m, n)
    elif kind == 2:    # Gram top-k
        b, c, n = ri(1, 3), ri(1, 64), ri(1, 2500)
        k = ri(1, min(64, n))
        x = torch.randn(b, c, n, generator=g).to(dev)
        dot = torch.matmul(x.transpose(2, 1), x).contiguous()
        sq = (x * x).sum(1).contiguous()
        got = gram_topk(dot, sq, k).long()
        val = (-sq.unsqueeze(1) - (-2 * dot)) - sq.unsqueeze(2)       # val[i][j] = (-sq[j] + 2 dot[i][j]) - sq[i]
        wv, wi = val.topk(k=k, dim=-1)
        assert torch.equal(torch.gather(val, 2, got), wv), ("topk values", b, c, n, k)
    elif kind == 3:    # shared-weight aggregation
        b, share, cw, k, n = ri(1, 3), [1, 2, 4, 8, 16][ri(0, 4)], ri(1, 9), ri(1, 24), ri(1, 3500)
        w = torch.randn(b, cw, k, n, generator=g).to(dev).requires_grad_()
        v = torch.randn(b, share * cw, k, n, generator=g).to(dev).requires_grad_()
        go = torch.randn(b, share * cw, n, generator=g).to(dev)
        out = share_weighted_sum(w, v)
        ref = (w.repeat(1, share, 1, 1) * v).sum(2)
        assert torch.allclose(out, ref, rtol=1e-4, atol=1e-4), ("aggregate", b, share, cw, k, n)
        for a_, b_ in zip(torch.autograd.grad(out, (w, v), go), torch.autograd.grad(ref, (w, v), go)):
            assert torch.allclose(a_, b_, rtol=1e-4, atol=1e-4), ("aggregate grad", b, share, cw, k, n)
    else:              # chamfer backward (LDS path below 8192 points in total)
        b, n, m = ri(1, 4), ri(1, 5000), ri(1, 5000)
        x1 = torch.rand(b, n, 3, generator=g).to(dev).requires_grad_()
        x2 = torch.rand(b, m, 3, generator=g).to(dev).requires_grad_()
        d1, d2, i1, i2 = cd()(x1, x2)
        g1, g2 = torch.rand(b, n, generator=g).to(dev), torch.rand(b, m, generator=g).to(dev)
        got = torch.autograd.grad((d1 * g1).sum() + (d2 * g2).sum(), (x1, x2))
        r1 = ((x1 - torch.gather(x2, 1, i1.long().unsqueeze(2).expand(-1, -1, 3))) ** 2).sum(2)
        r2 = ((x2 - torch.gather(x1, 1, i2.long().unsqueeze(2).expand(-1, -1, 3))) ** 2).sum(2)
        want = torch.autograd.grad((r1 * g1).sum() + (r2 * g2).sum(), (x1, x2))
        for a_, b_ in zip(got, want):
            assert torch.allclose(a_, b_, rtol=1e-4, atol=1e-4 * (1 + max(n, m) / min(n, m))), ("cd backward", b, n, m)
    cases += 1
torch.cuda.synchronize()
print("fuzz: %d cases in %.0f s, all consistent" % (cases, time.time() - t0))
