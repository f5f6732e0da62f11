"""Thread-level emulator of the reference's CUDA grids -- TEST INFRASTRUCTURE, second opinion.

`mvp_oracle.c` restates every reference kernel as sequential C and fixes the reference's one racy
step (GetMax's last writer) to one legal schedule.  It has a single author, and no reference-held
vectors exist for EMD / FPS.  This module is written independently of it, from the `.cu` text
alone, and keeps what the C restatement flattens away: the launch grid (`blockIdx`, `threadIdx`),
the shared-memory phases between `__syncthreads()`, the scan trees, the atomics -- every kernel is
a loop over blocks and threads that executes the kernel body per thread, phase by phase, with the
ORDER in which the threads of a phase run chosen by the caller:

    "ascending"   thread 0 first ... the last writer of a racy store is the highest thread id
    "descending"  the reverse
    an np.random.Generator: a fresh random permutation for every phase of every launch

A data race in the reference is therefore visible as a result that depends on the order; the set
of results over all orders is the reference program's outcome set.  The tests use it to show
(i) oracle == emulator under "ascending" (and oracle(getmax_lowest) == emulator under
"descending"), (ii) the HIP result equals the emulator's "ascending" member of that set, and
(iii) which launches are order-sensitive at all (GetMax with increments inside its 1e-6 band;
nothing else).

Only float32 arithmetic with fused multiply-adds is delegated (oracle/emu_arith.c, a dozen lines);
searches, merges, trees, counters and schedules are executed here.

Reference text followed: utils/metrics/EMD/emd_cuda.cu:23-226 (clear, calc_unass_cnt,
calc_unass_cnt_sum, calc_unass_idx, Bid, GetMax, Assign, CalcDist), emd_module.py:54-65 (initial
state), utils/mm3d_pn2/ops/furthest_point_sample/src/furthest_point_sample_cuda.cu:11-141.
Sizes: meant for n <= 2048 (seconds per call); it is a checker of checkers, not a baseline.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmvp_emu.so")
_lib = None
_f32p = ctypes.POINTER(ctypes.c_float)


def _arith():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "emu_arith.c")
        if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-B", "libmvp_emu.so"], stdout=subprocess.DEVNULL)
        _lib = ctypes.CDLL(_LIB)
        _lib.emu_calc_dist.restype = ctypes.c_float
    return _lib


def _p(a):
    return a.ctypes.data_as(_f32p)


class _Order:
    """The order in which the threads of one phase execute."""

    def __init__(self, schedule):
        self.schedule = schedule

    def __call__(self, count):
        if isinstance(self.schedule, np.random.Generator):
            return [int(i) for i in self.schedule.permutation(count)]
        if self.schedule == "ascending":
            return range(count)
        if self.schedule == "descending":
            return range(count - 1, -1, -1)
        raise ValueError(self.schedule)


# --------------------------------------------------------------------------------------- EMD
class EmdGrid:
    """One cloud pair's auction state and the reference's seven kernels per round (the batch
    dimension of the reference grids, blockIdx.x, only selects the cloud: clouds never interact,
    so one instance emulates gridDim.x = 1, b = 1).  All arrays are the reference's
    (emd_module.py:54-65), with its initial values."""

    BLOCK = 1024

    def __init__(self, xyz1, xyz2, eps, schedule="ascending"):
        self.xyz1 = np.ascontiguousarray(xyz1, dtype=np.float32)
        self.xyz2 = np.ascontiguousarray(xyz2, dtype=np.float32)
        n = self.n = self.xyz1.shape[0]
        assert self.xyz2.shape[0] == n and n % 1024 == 0          # emd_cuda.cu:236-249
        self.eps = np.float32(eps)
        self.order = _Order(schedule)
        self.assignment = np.full(n, -1, np.int32)
        self.assignment_inv = np.full(n, -1, np.int32)
        self.price = np.zeros(n, np.float32)
        self.bid = np.zeros(n, np.int32)
        self.bid_increments = np.zeros(n, np.float32)
        self.max_increments = np.zeros(n, np.float32)             # zeros, not -1e9 (emd_module.py:60)
        self.unass_idx = np.zeros(n, np.int32)
        self.max_idx = np.zeros(n, np.int32)
        self.unass_cnt = np.zeros(512, np.int32)
        self.unass_cnt_sum = np.zeros(512, np.int32)
        self.cnt_tmp = np.zeros(512, np.int32)
        self.racy_launches = 0       # GetMax launches in which two threads stored to one max_idx word

    # clear<<<1, batch_size>>> (:23-28)
    def k_clear(self):
        for t in self.order(1):      # blockDim.x = b = 1
            for i in range(t, 1, 1):
                self.cnt_tmp[i] = 0
                self.unass_cnt[i] = 0

    # calc_unass_cnt<<<(b, n/1024), 1024>>> (:30-55): up-sweep of a scan tree per block, atomicAdd of its root
    def k_calc_unass_cnt(self):
        B = self.BLOCK
        for by in self.order(self.n // B):
            scan = np.zeros(B, np.int64)
            for t in self.order(B):
                scan[t] = 1 if self.assignment[by * B + t] == -1 else 0
            # __syncthreads()
            stride = 1
            while stride <= B // 2:
                # every thread reads [index - stride] of the PREVIOUS level and writes [index] of this one:
                # disjoint within a level, so any order gives the same array
                for t in self.order(B):
                    index = (t + 1) * stride * 2 - 1
                    if index < B:
                        scan[index] += scan[index - stride]
                stride *= 2
                # __syncthreads()
            for t in self.order(B):
                if t == B - 1:
                    self.unass_cnt[0] += scan[t]                  # atomicAdd: commutative

    # calc_unass_cnt_sum<<<1, batch_size>>> (:57-86): inclusive scan over the batch (b = 1: the tree is empty)
    def k_calc_unass_cnt_sum(self):
        BS = 512
        scan = np.zeros(BS, np.int64)
        scan[0] = self.unass_cnt[0]          # threadIdx.x < batch_size = 1
        stride = 1
        while stride <= BS // 2:
            for t in self.order(1):
                index = (t + 1) * stride * 2 - 1
                if index < BS:
                    scan[index] += scan[index - stride]
            stride *= 2
        stride = BS // 4
        while stride > 0:
            for t in self.order(1):
                index = (t + 1) * stride * 2 - 1
                if index + stride < BS:
                    scan[index + stride] += scan[index]
            stride //= 2
        self.unass_cnt_sum[0] = scan[0]

    # calc_unass_idx<<<(b, n/1024), 1024>>> (:88-97): slot = atomicAdd(&cnt_tmp) -- the ORDER of the list
    # is whatever the hardware scheduled; nothing downstream depends on it (checked by the tests)
    def k_calc_unass_idx(self):
        B = self.BLOCK
        for by in self.order(self.n // B):
            for t in self.order(B):
                if self.assignment[by * B + t] == -1:
                    idx = int(self.cnt_tmp[0])
                    self.cnt_tmp[0] += 1
                    self.unass_idx[self.unass_cnt_sum[0] - self.unass_cnt[0] + idx] = by * B + t

    # Bid<<<(b, n/1024), 1024>>> (:99-179)
    def k_bid(self):
        n, B = self.n, self.BLOCK
        batch, block_size, block_cnt = 2048, 1024, n // 1024
        lib = _arith()
        _unass_cnt = int(self.unass_cnt[0])
        if _unass_cnt == 0:
            return
        _unass_cnt_sum = int(self.unass_cnt_sum[0])
        unass_per_block = (_unass_cnt + block_cnt - 1) // block_cnt
        thread_per_unass = block_size // unass_per_block
        scratch = np.empty(batch, np.float32)
        for by in self.order(block_cnt):
            unass_this_block = max(min(_unass_cnt - by * unass_per_block, unass_per_block), 0)
            # per-thread registers
            best = np.full(B, -1e9, np.float32)
            better = np.full(B, -1e9, np.float32)
            best_i = np.full(B, -1, np.int64)
            unass_id = np.full(B, -1, np.int64)
            thread_in_unass = np.zeros(B, np.int64)
            q = np.zeros((B, 3), np.float32)
            for t in self.order(B):
                if t < thread_per_unass * unass_this_block:
                    u = unass_per_block * by + t // thread_per_unass + _unass_cnt_sum - _unass_cnt
                    unass_id[t] = self.unass_idx[u]
                    thread_in_unass[t] = t % thread_per_unass
                    q[t] = self.xyz1[unass_id[t]]
            for k2 in range(0, n, batch):
                end_k = min(n, k2 + batch) - k2
                # cooperative tile load (disjoint stores), then __syncthreads()
                xyz2_buf = np.ascontiguousarray(self.xyz2[k2:k2 + end_k]).reshape(-1)
                price_buf = np.ascontiguousarray(self.price[k2:k2 + end_k])
                for t in self.order(B):
                    if unass_id[t] == -1:
                        continue
                    delta = (end_k + thread_per_unass - 1) // thread_per_unass
                    lo = int(thread_in_unass[t]) * delta
                    hi = min((int(thread_in_unass[t]) + 1) * delta, end_k)
                    if hi <= lo:
                        continue
                    lib.emu_bid_values(lo, hi, ctypes.c_float(q[t, 0]), ctypes.c_float(q[t, 1]), ctypes.c_float(q[t, 2]),
                                       _p(xyz2_buf), _p(price_buf), _p(scratch))
                    b_, bt_, bi_ = best[t], better[t], best_i[t]
                    for k in range(lo, hi):            # the thread's serial scan, strict comparisons (:147-154)
                        d = scratch[k - lo]
                        if d > b_:
                            bt_ = b_
                            b_ = d
                            bi_ = k + k2
                        elif d > bt_:
                            bt_ = d
                    best[t], better[t], best_i[t] = b_, bt_, bi_
                # __syncthreads()
            best_buf, better_buf, best_i_buf = best.copy(), better.copy(), best_i.copy()
            # __syncthreads(); the chunk leaders merge their followers' results serially (:163-171)
            for t in self.order(B):
                if unass_id[t] != -1 and thread_in_unass[t] == 0:
                    b_, bt_, bi_ = best[t], better[t], best_i[t]
                    for j in range(t + 1, t + thread_per_unass):
                        if best_buf[j] > b_:
                            bt_ = max(b_, better_buf[j])
                            b_ = best_buf[j]
                            bi_ = best_i_buf[j]
                        else:
                            bt_ = max(bt_, best_buf[j])
                    person = int(unass_id[t])
                    inc = np.float32(np.float32(b_ - bt_) + self.eps)
                    self.bid[person] = bi_
                    self.bid_increments[person] = inc
                    # atomicMax(float) by CAS (:9-20): the maximum, whatever the order
                    if inc > self.max_increments[bi_]:
                        self.max_increments[bi_] = inc

    # GetMax<<<(b, n/1024), 1024>>> (:181-194): plain stores to max_idx -- THE race of the reference
    def k_get_max(self):
        B = self.BLOCK
        writers = {}
        for by in self.order(self.n // B):
            for t in self.order(B):
                j = t + by * B
                if self.assignment[j] == -1:
                    bid_id = int(self.bid[j])
                    bid_inc = float(self.bid_increments[j])           # float -> double in `bid_inc - 1e-6`
                    max_inc = float(self.max_increments[bid_id])
                    if bid_inc - 1e-6 <= max_inc and max_inc <= bid_inc + 1e-6:
                        self.max_idx[bid_id] = j
                        writers[bid_id] = writers.get(bid_id, 0) + 1
        if any(c > 1 for c in writers.values()):
            self.racy_launches += 1

    # Assign<<<(b, n/1024), 1024>>> (:196-215)
    def k_assign(self, last):
        B = self.BLOCK
        for by in self.order(self.n // B):
            for t in self.order(B):
                j = t + by * B
                if self.assignment[j] == -1:
                    bid_id = int(self.bid[j])
                    if last or self.max_idx[bid_id] == j:
                        bid_inc = self.bid_increments[j]
                        ass_inv = int(self.assignment_inv[bid_id])
                        if not last and ass_inv != -1:
                            self.assignment[ass_inv] = -1
                        self.assignment_inv[bid_id] = j
                        self.assignment[j] = bid_id
                        self.price[bid_id] = np.float32(self.price[bid_id] + bid_inc)
                        self.max_increments[bid_id] = np.float32(-1e9)

    # Note on Assign: a thread tests `assignment[j] == -1` and another thread of the same launch may
    # store -1 to assignment[ass_inv].  ass_inv is an OWNER, j a bidder (unassigned at launch): the two
    # never name the same person, so the launch is order-independent -- except in the forced last
    # round, where several bidders may take the same object (assignment_inv / price: last writer, but
    # neither is read again; assignment[j] = bid[j] for every bidder whatever the order).

    def k_calc_dist(self):
        lib = _arith()
        dist = np.zeros(self.n, np.float32)
        for j in range(self.n):
            k = int(self.assignment[j])
            dist[j] = lib.emu_calc_dist(_p(self.xyz1[j]), _p(self.xyz2[k]))
        return dist

    def round(self, last):
        self.k_clear()
        self.k_calc_unass_cnt()
        self.k_calc_unass_cnt_sum()
        self.k_calc_unass_idx()
        self.k_bid()
        self.k_get_max()
        self.k_assign(last)


def emd_forward(xyz1, xyz2, eps, iters, schedule="ascending", return_info=False):
    """(B, n, 3) x 2 -> dist (B, n) float32, assignment (B, n) int32: emd_cuda_forward (:228-282)."""
    xyz1 = np.asarray(xyz1, np.float32)
    xyz2 = np.asarray(xyz2, np.float32)
    b, n, _ = xyz1.shape
    dist = np.zeros((b, n), np.float32)
    ass = np.zeros((b, n), np.int32)
    info = []
    for i in range(b):
        g = EmdGrid(xyz1[i], xyz2[i], eps, schedule)
        unassigned = []
        for it in range(iters):
            unassigned.append(int((g.assignment == -1).sum()))
            g.round(it == iters - 1)
        dist[i] = g.k_calc_dist()
        ass[i] = g.assignment
        info.append({"racy_getmax_launches": g.racy_launches, "unassigned": unassigned})
    return (dist, ass, info) if return_info else (dist, ass)


# --------------------------------------------------------------------------------------- FPS
def opt_n_threads(work_size):
    """furthest_point_sample_cuda.cu:11-15 (float log, as written)."""
    pow_2 = int(math.log(float(work_size)) / math.log(2.0))
    return max(min(1 << pow_2, 1024), 1)


def furthest_point_sample(xyz, m, schedule="ascending"):
    """furthest_point_sampling_kernel<block_size> (:26-141) for every cloud of xyz (B, N, 3) ->
    idx (B, m) int32: per-thread strided scan with strict `>` (first maximum in k order), then the
    shared-memory tree of __update(v2 > v1 ? i2 : i1), level by level; `temp` starts at 1e10
    (furthest_point_sample.py:30).  No step of it is racy: the schedule argument exists to show that."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    b, n, _ = xyz.shape
    lib = _arith()
    order = _Order(schedule)
    out = np.zeros((b, m), np.int32)
    for c in range(b):
        pts = xyz[c]
        bs = opt_n_threads(n)
        temp = np.full(n, 1e10, np.float32)
        old = 0
        out[c, 0] = old
        d = np.empty((n + bs - 1) // bs, np.float32)
        for j in range(1, m):
            dists = np.zeros(bs, np.float32)
            dists_i = np.zeros(bs, np.int64)
            x1, y1, z1 = pts[old]
            for tid in order(bs):
                besti, best = 0, np.float32(-1)
                cnt = len(range(tid, n, bs))
                if cnt:
                    lib.emu_fps_sqdist(int(tid), bs, n, ctypes.c_float(x1), ctypes.c_float(y1), ctypes.c_float(z1), _p(pts), _p(d))
                for q, k in enumerate(range(tid, n, bs)):
                    d2 = min(d[q], temp[k])
                    temp[k] = d2
                    if d2 > best:
                        besti, best = k, d2
                dists[tid], dists_i[tid] = best, besti
            # __syncthreads(); tree: block_size >= 2*s  ->  tid < s: __update(tid, tid + s)
            s = bs // 2
            while s >= 1:
                for tid in order(s):             # reads [tid], [tid + s], writes [tid]: disjoint per level
                    v1, v2 = dists[tid], dists[tid + s]
                    i1, i2 = dists_i[tid], dists_i[tid + s]
                    dists[tid] = max(v1, v2)
                    dists_i[tid] = i2 if v2 > v1 else i1
                s //= 2
            old = int(dists_i[0])
            out[c, j] = old
    return out
