"""Auction EMD operator -- same surface as the reference's
utils/metrics/EMD/emd_module.py (emdFunction :40-81, emdModule :83-88),
backed by libmvpops' persistent auction kernel (mvp_emd_forward).

Input:  xyz1 = predicted cloud, xyz2 = ground truth, both (B, n, 3) in [0, 1];
        n a multiple of 1024, B <= 512 (emd_cuda.cu:236-249); eps, iters.
Output: dist (B, n) squared matched distances (sqrt -> L2), assignment (B, n)
        int32 indices into xyz2 (not guaranteed to be a bijection).
Gradient flows to xyz1 only (emd_module.py:73-81).

The reference allocates 11 scratch tensors per call (:54-65); here one byte
buffer of mvp_emd_scratch_bytes(B, n) is enough and its initial contents do
not matter.  Shape guards raise instead of printf + ignored return code.

Failure contract: if the kernel abandons a cluster wait (never observed; needs
the workgroups of one cloud not to be co-resident for tens of seconds) or trips
an internal check, the cloud's status word is negative, its `dist` NaN and its
`assignment` -1 (abandoned wait), so every metric derived from it is NaN instead
of silently wrong; backward skips such entries (zero gradient).  The status
words are ALWAYS checked, without a host synchronisation: every forward copies
them to pinned host memory behind its kernels and the NEXT forward -- or an
explicit `emd_module.check()` (blocking; the eval loop calls it once at the
end) -- raises MvpOpsError for any call that failed (`LAZY_STATUS = False`
switches that off).  `CHECK_STATUS = True` checks synchronously in the same call.
"""
import threading

import torch
from torch import nn
from torch.autograd import Function

from ..._lib import MvpOpsError, call, emd_scratch_bytes

# True: every forward reads the kernel's per-cloud status words back at once (a host
# synchronisation) and raises MvpOpsError on a failed cloud.
CHECK_STATUS = False
# True (default): the status words of every call are copied to pinned host memory
# asynchronously and examined by the next forward / by check() -- a failure is loud
# one call late, at no synchronisation cost.
LAZY_STATUS = True
_PENDING = []          # (event, pinned int64 rounds, description) of calls not examined yet
_PINNED_POOL = []      # recycled pinned buffers
_LOCK = threading.RLock()   # both lists: forward may run on several threads (nn.DataParallel, one per GPU)


def _examine(rounds, what):
    bad = rounds[rounds < 0]
    if bad.numel():
        raise MvpOpsError("mvp_emd_forward failed for %d cloud(s) of an earlier call (%s): status %s"
                          % (bad.numel(), what, bad.tolist()))


def _capturing(device=None):
    """True while `device`'s current stream is being captured into a graph: event queries, pinned copies
    and the rest of this bookkeeping are not graph work (and an event query would invalidate the capture)."""
    if not torch.cuda.is_available():
        return False
    if device is None:
        return torch.cuda.is_current_stream_capturing()
    with torch.cuda.device(device):
        return torch.cuda.is_current_stream_capturing()


def check(block=True):
    """Examine the status words of the EMD calls made so far (block=True: wait for them).
    A no-op while a stream is being captured (nothing may be queried then)."""
    if _capturing():
        return
    with _LOCK:
        keep = []
        try:
            while _PENDING:
                ev, host, what = _PENDING.pop(0)
                if block:
                    ev.synchronize()
                elif not ev.query():
                    keep.append((ev, host, what))
                    continue
                try:
                    _examine(host, what)
                finally:
                    if len(_PINNED_POOL) < 8:
                        _PINNED_POOL.append(host)
        finally:
            _PENDING[:0] = keep


def _file_status(scratch, nbytes, batchsize, n, eps, iters):
    """Copy the call's status words to pinned memory behind its kernels and file an event that says when
    they have landed.  Copy and event belong to the stream the kernels were launched on: the current stream
    of the TENSORS' device, which need not be the current device (nn.DataParallel drives several)."""
    device = scratch.device
    if _capturing(device):
        return              # (a captured graph replays the launch, not this bookkeeping)
    rounds = scratch[nbytes - batchsize * 16:].view(torch.int64).view(batchsize, 2)[:, 0]
    with _LOCK:
        host = None
        for i, h in enumerate(_PINNED_POOL):
            if h.numel() == batchsize:
                host = _PINNED_POOL.pop(i)
                break
    if host is None:
        host = torch.empty(batchsize, dtype=torch.int64, pin_memory=True)
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream(device)
        with torch.cuda.stream(stream):
            host.copy_(rounds, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(stream)
    with _LOCK:
        _PENDING.append((ev, host, "b=%d n=%d eps=%g iters=%d on %s" % (batchsize, n, eps, iters, device)))
        overfull = len(_PENDING) > 64
    if overfull:      # bounded: a loop that never reaches check() still examines the old ones
        check(block=False)
        with _LOCK:
            overfull = len(_PENDING) > 64
        if overfull:
            check(block=True)


class emdFunction(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2, eps, iters):
        batchsize, n, _ = xyz1.size()
        _, m, _ = xyz2.size()

        assert n == m
        assert xyz1.size()[0] == xyz2.size()[0]
        assert batchsize <= 512
        if n % 1024 != 0:
            raise ValueError("Input Error! The size of the point clouds should "
                             "be a multiple of 1024.")

        xyz1 = xyz1.contiguous().float()
        xyz2 = xyz2.contiguous().float()
        device = xyz1.device
        if LAZY_STATUS and _PENDING and not _capturing(device):
            check(block=False)
        dist = torch.zeros(batchsize, n, device=device)
        assignment = torch.zeros(batchsize, n, device=device,
                                 dtype=torch.int32) - 1
        nbytes = emd_scratch_bytes(batchsize, n)
        scratch = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device)

        call("mvp_emd_forward", device, batchsize, n, xyz1, xyz2, dist,
             assignment, eps, iters, scratch, nbytes)

        if CHECK_STATUS:   # debug aid: one device->host read per call
            rounds = scratch[nbytes - batchsize * 16:].view(torch.int64).view(batchsize, 2)[:, 0]
            if bool((rounds < 0).any()):
                raise MvpOpsError("mvp_emd_forward abandoned %d cloud(s) (status %s)"
                                  % (int((rounds < 0).sum()), rounds[rounds < 0].tolist()))
        elif LAZY_STATUS:
            _file_status(scratch, nbytes, batchsize, n, eps, iters)

        ctx.save_for_backward(xyz1, xyz2, assignment)
        ctx.mark_non_differentiable(assignment)
        return dist, assignment

    @staticmethod
    def backward(ctx, graddist, gradidx):
        xyz1, xyz2, assignment = ctx.saved_tensors
        graddist = graddist.contiguous()
        batchsize, n, _ = xyz1.size()

        gradxyz1 = torch.zeros(xyz1.size(), device=xyz1.device)
        gradxyz2 = torch.zeros(xyz2.size(), device=xyz2.device)
        call("mvp_emd_backward", xyz1.device, batchsize, n, xyz1, xyz2,
             gradxyz1, graddist, assignment)
        return gradxyz1, gradxyz2, None, None


class emdModule(nn.Module):
    def __init__(self):
        super(emdModule, self).__init__()

    def forward(self, input1, input2, eps, iters):
        return emdFunction.apply(input1, input2, eps, iters)


def test_emd(batch=20, n=8192, eps=0.05, iters=3000, device='cuda'):
    """Smoke run in the spirit of the reference's test_emd (:90-104); returns
    (mean sqrt(dist), distinct targets, mean sqrt of the distance recomputed
    from the returned assignment) instead of printing."""
    x1 = torch.rand(batch, n, 3, device=device)
    x2 = torch.rand(batch, n, 3, device=device)
    dis, assignment = emdModule()(x1, x2, eps, iters)
    matched = torch.gather(x2, 1, assignment.long().unsqueeze(-1).expand(-1, -1, 3))
    verified = ((x1 - matched) ** 2).sum(-1).sqrt().mean()
    return dis.sqrt().mean().item(), assignment.unique().numel(), verified.item()
