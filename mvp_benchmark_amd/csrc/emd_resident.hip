// LDS-resident tail of the EMD auction for gfx950: clouds of at most 4096 points.
//
// Once a cloud of n <= 4096 points has at most kResCap unassigned persons (round ~30 of 3000 at
// n = 1024, ~100 at 2048, ~200-350 at 4096; their number never grows) the WHOLE auction state fits one
// CU's 160 KB of LDS: objects {x, y, z, price} 16 B, owner 2 B, the persons' points 12 B and bid hints
// 4 B per point.  emd_lean.hip's kernels stop such a cloud (hand-over record + lists, exactly as the
// first kernel leaves them for the lean kernel) and emd_resident_kernel runs the remaining rounds --
// 90-99 % of them -- on ONE workgroup per cloud with no global memory access in the round at all:
// a round is LDS reads, two workgroup barriers and nothing else (clustered kernels: three dependent
// L2 round trips per bid, two cluster all-gathers and Assign's two trips per round, ~5 us; here ~1 us).
// Same rounds, same bids, same bits (utils/metrics/EMD/emd_cuda.cu:95-215; tests/test_gpu_ops.py):
//   * Bid (emd_cuda.cu:95-179): one wave per bidder.  Seed = the exact values of the bidder's previous
//     best / second-best objects (two distinct real objects: a valid lower bound B2 of the second-best
//     value); the objects are cell-sorted, so 64 consecutive slots are a spatially compact block with an
//     exact bounding box and a lower bound of its prices: one lane per block tests
//     dist(q, box) + price_lb <= (3 - B2) + margin, the surviving blocks are scanned 64 objects per step
//     with the clustered kernels' lossless filter (emd_common.h: kMargin) and exact values are folded in
//     the reference's tie order (emd_fold);
//   * GetMax + Assign (emd_cuda.cu:181-215): the <= 64 bids of the round sit in the lanes of wave 0; the
//     winner of an object is the highest bidder index within 1e-6 of its maximal increment (what the
//     reference's racy GetMax gives when executed for ascending thread ids -- the schedule the oracle
//     pins), found by comparing the bids with each other: no keys, no atomics;
//   * the other waves meanwhile refresh the blocks' price bounds (prices only rise: a bound read while a
//     price is being raised is still a bound).
#include "emd_common.h"

namespace mvp {

// (kResList, kResMaxN: emd_common.h)
constexpr int kResBlock = 64;   // slots per block = lanes per wave
constexpr unsigned short kResFree = 0xFFFFu;

template <int NMAX>
struct ResShared {
  float4 obj[NMAX];                      // slot -> x, y, z, price
  float px[NMAX], py[NMAX], pz[NMAX];    // person -> point
  unsigned short owner[NMAX];            // slot -> person (kResFree: none)
  unsigned short h1[NMAX], h2[NMAX];     // person -> best / second-best slot of its last bid (seed hints)
  float4 b_lo[NMAX / kResBlock], b_hi[NMAX / kResBlock];   // block: box min + price lower bound / box max
  int list[2][kResList];                 // unassigned persons of this / the next round
  int s_bj[kResList], s_bo[kResList];
  float s_binc[kResList];
  int s_cnt[2];
  int s_err;
  int s_next;
};

template <int NMAX>
__global__ __launch_bounds__(kEmdThreads) void emd_resident_kernel(
    int b, int n, float *__restrict__ dist, int *assignment, float eps, int iters, char *scratch) {
  __shared__ ResShared<NMAX> sh;
  const int cloud = (int)blockIdx.x;
  if (cloud >= b) return;
  const int t = threadIdx.x;
  const int lane = t & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // (tells the compiler that it is wave-uniform)
  char *tail = scratch + (size_t)b * emd_scratch_per_cloud(n);
  EmdHandover *resume = emd_handover(tail, b, cloud);
  long long *stats = emd_stats(tail, b, cloud);
  const int it0 = resume->next_it;
  if (it0 == 0) return;   // finished in an earlier launch (uniform over the workgroup)
  const EmdScratch sc = emd_carve(scratch + (size_t)cloud * emd_scratch_per_cloud(n), n);
  dist += (size_t)cloud * n;
  int *ass = assignment + (size_t)cloud * n;
  const int nblk = n / kResBlock;   // n % 1024 == 0

  // ------------------------------------------------------------ load the auction state
  for (int s = t; s < n; s += kEmdThreads) {
    sh.obj[s] = sc.obj[s];
    const int ow = sc.ostate[s].z;
    sh.owner[s] = ow < 0 ? kResFree : (unsigned short)ow;
    const float4 pa = sc.person[2 * s], pb = sc.person[2 * s + 1];
    sh.px[s] = pa.x;
    sh.py[s] = pa.y;
    sh.pz[s] = pa.z;
    const int p1 = __float_as_int(pb.y), p2 = __float_as_int(pb.z);
    sh.h1[s] = p1 < 0 ? kResFree : (unsigned short)p1;
    sh.h2[s] = p2 < 0 ? kResFree : (unsigned short)p2;
  }
  {
    // the lists the previous launch left (its cluster width: nlists), concatenated
    const int nl = resume->nlists;
    int p = t, k = -1, total = 0;
#pragma unroll
    for (int w = 0; w < kMaxCluster; ++w) {
      const int cw = w < nl ? resume->cnt[w] : 0;
      if (k < 0 && p >= 0 && p < cw) k = sc.ulist[(size_t)w * 2 * n + p];
      p -= cw;
      total += cw;
    }
    if (t < total && t < kResList) sh.list[0][t] = k;
    if (t == 0) {
      sh.s_cnt[0] = min(total, kResList);
      sh.s_cnt[1] = 0;
      sh.s_err = (resume->err != 0 || total > kResList) ? 1 : 0;   // (the launcher never hands over more)
      sh.s_next = kEmdWaves;
    }
  }
  __syncthreads();
  // exact bounding box and exact price minimum per block of 64 slots
  for (int blk = wave; blk < nblk; blk += kEmdWaves) {
    const float4 o = sh.obj[blk * kResBlock + lane];
    float lx = o.x, ly = o.y, lz = o.z, lw = o.w, hx = o.x, hy = o.y, hz = o.z;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      lx = __builtin_fminf(lx, __shfl_xor(lx, off, kWave));
      ly = __builtin_fminf(ly, __shfl_xor(ly, off, kWave));
      lz = __builtin_fminf(lz, __shfl_xor(lz, off, kWave));
      lw = __builtin_fminf(lw, __shfl_xor(lw, off, kWave));
      hx = __builtin_fmaxf(hx, __shfl_xor(hx, off, kWave));
      hy = __builtin_fmaxf(hy, __shfl_xor(hy, off, kWave));
      hz = __builtin_fmaxf(hz, __shfl_xor(hz, off, kWave));
    }
    if (lane == 0) {
      sh.b_lo[blk] = make_float4(lx, ly, lz, lw);
      sh.b_hi[blk] = make_float4(hx, hy, hz, 0.f);
    }
  }
  __syncthreads();

  // ------------------------------------------------------------ the auction
  int cur = 0;
  long long n_rounds = 0, n_bids = 0;
  int U = __builtin_amdgcn_readfirstlane(sh.s_cnt[0]);
  int last_u = 0;
#ifdef MVP_EMD_PROFILE
  long long prof_folds = 0, prof_blocks = 0, prof_bidcyc = 0, prof_nbid = 0, cyc_bid = 0, cyc_sync1 = 0, cyc_assign = 0;
  const long long t_loop0 = __builtin_readcyclecounter();
#endif   // bidders of the forced last round (their bids are their assignment)
  for (int it = it0; it < iters; ++it) {
    if (U == 0) break;
    n_rounds += 1;
    n_bids += U;
    const bool last = it == iters - 1;
    const int tpu = -U;   // thread_per_unass (emd_cuda.cu:107-109), resolved inside emd_precedes: ties only

    // ---------------- Bid (emd_cuda.cu:95-179): one wave per bidder; positions beyond the first 16 are drawn
#ifdef MVP_EMD_PROFILE
    const long long tp0 = __builtin_readcyclecounter();
#endif
    int u = wave;
    for (int guard = 0; guard <= kResList && u < U; ++guard) {
      const int j = __builtin_amdgcn_readfirstlane(sh.list[cur][u]);
      const float qx = sh.px[j], qy = sh.py[j], qz = sh.pz[j];
      int p1 = __builtin_amdgcn_readfirstlane((int)sh.h1[j]);
      if (__builtin_expect(p1 == kResFree, 0)) p1 = 0;   // (every person has bid before a hand-over; any block gives a valid seed)
      BidState st;
      st.b1 = -1e9f;
      st.b2 = -1e9f;
      st.bk = -1;
      st.b2k = -1;
#ifdef MVP_EMD_PROFILE
      const long long tb0 = __builtin_readcyclecounter();
#endif
      {
        // seed: the second-largest exact value among the 64 objects of the block that holds the bidder's
        // previous best object -- real, distinct objects: a valid lower bound of the final second-best value
        const float4 o = sh.obj[(p1 & ~(kResBlock - 1)) + lane];
        const float v = emd_value(sqdist3(o.x - qx, o.y - qy, o.z - qz), o.w);
        float t1, t2;
        wave_top2(v, t1, t2);
        st.tm = (3.0f - t2) + kMargin;
      }
      // one lane per block: squared distance to the block's box, its price bound
      float bd2 = __builtin_inff(), bpl = 0.f;
      if (lane < nblk) {
        const float4 lo = sh.b_lo[lane], hi = sh.b_hi[lane];
        const float dx = __builtin_fmaxf(__builtin_fmaxf(lo.x - qx, qx - hi.x), 0.f);
        const float dy = __builtin_fmaxf(__builtin_fmaxf(lo.y - qy, qy - hi.y), 0.f);
        const float dz = __builtin_fmaxf(__builtin_fmaxf(lo.z - qz, qz - hi.z), 0.f);
        bd2 = sqdist3(dx, dy, dz);
        bpl = lo.w;
      }
      auto blocks = [&]() -> unsigned long long {
        const float tq = st.tm - bpl;
        return __ballot(tq >= 0.f && bd2 <= tq * tq);   // (lanes >= nblk: inf)
      };
      unsigned long long bm = blocks();
      // the block of the previous best object first: it usually holds today's best as well, and the
      // threshold it leaves prunes the rest
      unsigned long long first = bm & (1ull << (p1 >> 6));
      while (bm) {
        int bi[4];
        unsigned long long take = bm;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (first) {
            bi[r] = (int)__builtin_ctzll(first);
            take &= ~first;
            first = 0ull;
          } else if (take) {
            bi[r] = (int)__builtin_ctzll(take);
            take &= take - 1;
          } else {
            bi[r] = -1;
          }
        }
        bm = take;
        float4 o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = sh.obj[(bi[r] < 0 ? 0 : bi[r]) * kResBlock + lane];
        bool tightened = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (bi[r] < 0) continue;   // (uniform)
          const float sd = sqdist3(o[r].x - qx, o[r].y - qy, o[r].z - qz);
          const float tq = st.tm - o[r].w;
          const unsigned long long m = __ballot(tq >= 0.f && sd <= tq * tq);
          if (m) {
            emd_fold(st, m, emd_value(sd, o[r].w), bi[r] * kResBlock + lane, n, tpu, sc.perm);
            tightened = true;
#ifdef MVP_EMD_PROFILE
            prof_folds += __builtin_popcountll(m);
#endif
          }
#ifdef MVP_EMD_PROFILE
          prof_blocks += 1;
#endif
        }
        if (tightened) bm &= blocks();
      }
      if (__builtin_expect(st.bk < 0 || st.b2k < 0, 0)) {   // cannot happen: the two seed objects pass the filter
        if (lane == 0) sh.s_err = 1;
        st.bk = st.bk < 0 ? 0 : st.bk;
        st.b2k = st.b2k < 0 ? (st.bk == 0 ? 1 : 0) : st.b2k;
      }
#ifdef MVP_EMD_PROFILE
      prof_bidcyc += __builtin_readcyclecounter() - tb0;
      prof_nbid += 1;
#endif
      int drawn = 0;
      if (lane == 0) {
        sh.s_bj[u] = j;
        sh.s_bo[u] = st.bk;
        sh.s_binc[u] = st.b1 - st.b2 + eps;
        sh.h1[j] = (unsigned short)st.bk;
        sh.h2[j] = (unsigned short)st.b2k;
        drawn = atomicAdd(&sh.s_next, 1);
      }
      u = __builtin_amdgcn_readlane(drawn, 0);
    }
#ifdef MVP_EMD_PROFILE
    const long long tp1 = __builtin_readcyclecounter();
#endif
    lds_barrier();
#ifdef MVP_EMD_PROFILE
    const long long tp2 = __builtin_readcyclecounter();
    cyc_bid += tp1 - tp0;
    cyc_sync1 += tp2 - tp1;
#endif

    // ---------------- GetMax + Assign (emd_cuda.cu:181-215): a lane per bid (waves 0 and 1)
    const int nxt = cur ^ 1;
    if (last) {
      last_u = U;   // every bidder of the last round takes what it bid on (emd_cuda.cu:201-212); resolved below the loop
      break;
    }
    if (t < kResList) {
      const bool act = t < U;
      const int j = act ? sh.s_bj[t] : -1, o = act ? sh.s_bo[t] : -1;
      const float inc = act ? sh.s_binc[t] : 0.f;
      // the maximal increment bid on my object, then the highest bidder inside its 1e-6 band
      float mi = inc;
      int wj = -1;
      if (__builtin_expect(U <= kWave, 1)) {
        // (the round's bids are the lanes of wave 0: compared through readlane, no LDS traffic)
        for (int v = 0; v < U; ++v) {
          const int ov = __builtin_amdgcn_readlane(o, v);
          const float iv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inc), v));
          if (ov == o) mi = __builtin_fmaxf(mi, iv);
        }
        for (int v = 0; v < U; ++v) {
          const int ov = __builtin_amdgcn_readlane(o, v), jv = __builtin_amdgcn_readlane(j, v);
          const float iv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inc), v));
          if (ov == o && emd_in_band(iv, mi)) wj = max(wj, jv);
        }
      } else {
        for (int v = 0; v < U; ++v)
          if (sh.s_bo[v] == o) mi = __builtin_fmaxf(mi, sh.s_binc[v]);
        for (int v = 0; v < U; ++v)
          if (sh.s_bo[v] == o && emd_in_band(sh.s_binc[v], mi)) wj = max(wj, sh.s_bj[v]);
      }
      if (act) {
        // next round's list: evicted owners and losers (the order of a list changes no result)
        int again = j;
        if (wj == j) {   // one winner per object
          again = sh.owner[o];
          again = again == kResFree ? -1 : again;
          sh.owner[o] = (unsigned short)j;
          sh.obj[o].w = sh.obj[o].w + inc;
        }
        if (again >= 0) sh.list[nxt][atomicAdd(&sh.s_cnt[nxt], 1)] = again;   // (<= U entries: never grows)
      }
    } else {
      // the other waves refresh block price bounds meanwhile.  A price read while wave 0 raises it is
      // the old or the new one; both are <= every later price: the minimum stays a lower bound.
      const int blk = (int)(((unsigned)it * (unsigned)(kEmdWaves - 2) + (unsigned)(wave - 2)) % (unsigned)nblk);
      float pm = sh.obj[blk * kResBlock + lane].w;
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) pm = __builtin_fminf(pm, __shfl_xor(pm, off, kWave));
      if (lane == 0) sh.b_lo[blk].w = pm;
    }
    if (t == 0) {
      sh.s_cnt[cur] = 0;        // the list after next
      sh.s_next = kEmdWaves;    // list positions 0..15 belong to the waves, the rest are drawn
    }
    lds_barrier();
    cur = nxt;
    U = __builtin_amdgcn_readfirstlane(sh.s_cnt[cur]);
#ifdef MVP_EMD_PROFILE
    cyc_assign += __builtin_readcyclecounter() - tp2;
#endif
  }
#ifdef MVP_EMD_PROFILE
  if (cloud < 2 && lane == 0 && (wave == 0 || wave == 5))
    printf("resident cloud %d wave %d: rounds %lld bids(all waves) %lld | this wave: %lld bids, %lld cycles each, blocks %.1f folds %.1f per bid | cycles bid %lld wait %lld assign %lld total %lld\n",
           cloud, wave, n_rounds, n_bids, prof_nbid, prof_bidcyc / (prof_nbid + 1), (double)prof_blocks / (double)(prof_nbid + 1),
           (double)prof_folds / (double)(prof_nbid + 1), cyc_bid, cyc_sync1, cyc_assign, __builtin_readcyclecounter() - t_loop0);
#endif

  // ------------------------------------------------------------ assignment + CalcDist (emd_cuda.cu:217-226)
  // person -> slot: what the owners say, then the last round's bids (the reference's last round
  // evicts nobody and gives every bidder the object it bid on: several persons may share one)
  __syncthreads();
  unsigned short *pslot = sh.h1;   // (the hints are not needed any more)
  for (int s = t; s < n; s += kEmdThreads) {
    const unsigned short ow = sh.owner[s];
    if (ow != kResFree) pslot[ow] = (unsigned short)s;
  }
  __syncthreads();
  if (t < last_u) pslot[sh.s_bj[t]] = (unsigned short)sh.s_bo[t];
  __syncthreads();
  for (int j = t; j < n; j += kEmdThreads) {
    const int s = pslot[j];
    const float4 o = sh.obj[s];
    dist[j] = sqdist3(sh.px[j] - o.x, sh.py[j] - o.y, sh.pz[j] - o.z);
    ass[j] = sc.perm[s];
  }
  if (t == 0) {
    atomicAdd(reinterpret_cast<unsigned long long *>(&stats[0]), (unsigned long long)n_rounds);
    atomicAdd(reinterpret_cast<unsigned long long *>(&stats[1]), (unsigned long long)n_bids);
    if (sh.s_err) atomicAdd(reinterpret_cast<unsigned long long *>(&stats[0]), (unsigned long long)(-(1ll << 40)));
    resume->next_it = 0;              // finished
    resume->last_width = 1 + 16 * 3;  // one workgroup, the resident launch
  }
}

// Finishes the clouds emd_lean.hip's launch stopped for it (hand-over records with next_it != 0).
hipError_t emd_resident_launch(int b, int n, float *dist, int *assignment, float eps, int iters, char *scratch,
                               hipStream_t stream) {
  if (n <= 2048)
    hipLaunchKernelGGL(emd_resident_kernel<2048>, dim3(b), dim3(kEmdThreads), 0, stream, b, n, dist, assignment, eps,
                       iters, scratch);
  else if (n <= kResMaxN)
    hipLaunchKernelGGL(emd_resident_kernel<4096>, dim3(b), dim3(kEmdThreads), 0, stream, b, n, dist, assignment, eps,
                       iters, scratch);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace mvp
