// LDS-resident tail of the EMD auction for gfx950: clouds of at most 4096 points.
//
// Once a cloud of n <= 4096 points has at most `res_cap` (<= 64) unassigned persons (round ~30 of 3000 at
// n = 1024, ~100 at 2048, 200-350 at 4096; their number never grows) the WHOLE auction state fits one
// CU's 160 KB of LDS: objects {x, y, z, price} 16 B, owner 2 B, the persons' points 12 B and a bid hint
// 2 B per point.  emd_lean.hip's kernels stop such a cloud (hand-over record + lists, exactly as the
// first kernel leaves them for the lean kernel) and emd_resident_kernel runs the remaining rounds --
// 90-99 % of them -- on ONE workgroup per cloud with no global memory access in the round at all
// (clustered kernels: three dependent L2 round trips per bid, two cluster all-gathers and Assign's two
// trips per round).  What bounds a round here is the length of its dependent instruction chain (a wave
// issues a dependent instruction every ~8 cycles, an LDS round trip is ~64), so a round is two phases of
// wave-private, mostly straight-line work and two workgroup barriers:
//   * Bid (emd_cuda.cu:95-179): one wave per bidder.  The objects are cell-sorted, so 64 consecutive slots
//     are a spatially compact block with an exact bounding box and a lower bound of its prices.  The block
//     that holds the bidder's previous best object is evaluated exactly first: its two best values start
//     the running top two, and the second of them is a lower bound B2 of the final second-best value; one
//     lane per block tests dist(q, box) + price_lb <= (3 - B2) + margin, the surviving blocks are scanned
//     four at a time, 64 objects each, with the clustered kernels' lossless filter (emd_common.h: kMargin),
//     and exact values are folded in the reference's tie order (emd_fold);
//   * GetMax + Assign (emd_cuda.cu:181-215): the wave that placed a bid also settles it.  A bid counts
//     itself in a 256-bucket table (by object slot) when it is placed; a count of one after the barrier
//     means nobody else bid on that object and the bidder wins without further ado (nearly always).
//     Otherwise the wave compares the round's bids (all in LDS): the winner of an object is the highest
//     bidder index within 1e-6 of its maximal increment -- what the reference's racy GetMax gives when
//     executed for ascending thread ids, the schedule the oracle pins.  No keys, no returning atomics on
//     the path, no single wave that serialises the round;
//   * waves without a bidder refresh the blocks' price bounds meanwhile (prices only rise: a stale bound
//     is still a bound).
// Same rounds, same bids, same bits as every other launch sequence (tests/test_gpu_emd_resident.py).
#include "emd_common.h"

namespace mvp {

// (kResList, kResMaxN, kResMinRounds: emd_common.h)
constexpr int kResBlock = 64;   // slots per block = lanes per wave
constexpr int kResBuckets = 256;
constexpr unsigned short kResFree = 0xFFFFu;

template <int NMAX>
struct ResShared {
  float4 obj[NMAX];                      // slot -> x, y, z, price
  float px[NMAX], py[NMAX], pz[NMAX];    // person -> point
  unsigned short owner[NMAX];            // slot -> person (kResFree: none)
  unsigned short h1[NMAX];               // person -> slot it last bid on (seed hint); at the end: person -> slot
  float4 b_lo[NMAX / kResBlock], b_hi[NMAX / kResBlock];   // block: box min + price lower bound / box max
  int cnt[2][kResBuckets];               // by round parity: bids per bucket of object slots
  int list[2][kResList];                 // unassigned persons of this / the next round
  int s_bj[kResList], s_bo[kResList];    // this round's bids: person, slot,
  float s_binc[kResList];                // increment
  int s_cnt[2];
  int s_err;
};

// min over the wave, valid in lane 63
__device__ __forceinline__ float wave_min_lane63(float v) {
  const float inf = __builtin_inff();
  v = __builtin_fminf(v, dpp_f32<0xB1, 0xF>(inf, v));    // quad_perm [1,0,3,2]
  v = __builtin_fminf(v, dpp_f32<0x4E, 0xF>(inf, v));    // quad_perm [2,3,0,1]
  v = __builtin_fminf(v, dpp_f32<0x141, 0xF>(inf, v));   // row_half_mirror
  v = __builtin_fminf(v, dpp_f32<0x140, 0xF>(inf, v));   // row_mirror
  v = __builtin_fminf(v, dpp_f32<0x142, 0xA>(inf, v));   // row_bcast15 -> rows 1, 3
  v = __builtin_fminf(v, dpp_f32<0x143, 0xC>(inf, v));   // row_bcast31 -> rows 2, 3
  return v;
}

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
    const int p1 = __float_as_int(pb.y);
    sh.h1[s] = p1 < 0 ? kResFree : (unsigned short)p1;
  }
  if (t < 2 * kResBuckets) (&sh.cnt[0][0])[t] = 0;
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
  int last_u = 0;   // bidders of the forced last round (their bids are their assignment)
#ifdef MVP_EMD_PROFILE
  long long prof_folds = 0, prof_blocks = 0, prof_bidcyc = 0, prof_nbid = 0, cyc_bid = 0, cyc_sync1 = 0, cyc_assign = 0, prof_slow = 0,
            prof_seed = 0;
  const long long t_loop0 = __builtin_readcyclecounter();
  const long long w_loop0 = wall_clock64();
#endif
  for (int it = it0; it < iters; ++it) {
    if (U == 0) break;
    n_rounds += 1;
    n_bids += U;
    const bool last = it == iters - 1;
    const int tpu = -U;   // thread_per_unass (emd_cuda.cu:107-109), resolved inside emd_precedes: ties only
    int *cnt = sh.cnt[it & 1];

    // ---------------- Bid (emd_cuda.cu:95-179): wave w bids for the list positions w, w + 16, ...
#ifdef MVP_EMD_PROFILE
    const long long tp0 = __builtin_readcyclecounter();
#endif
    for (int u = wave; u < U; u += kEmdWaves) {
      const int j = __builtin_amdgcn_readfirstlane(sh.list[cur][u]);
      const float qx = sh.px[j], qy = sh.py[j], qz = sh.pz[j];
      int p1 = __builtin_amdgcn_readfirstlane((int)sh.h1[j]);
      if (__builtin_expect(p1 == kResFree, 0)) p1 = 0;   // (every person has bid before a hand-over; any block will do)
      const int home = p1 >> 6;
#ifdef MVP_EMD_PROFILE
      const long long tb0 = __builtin_readcyclecounter();
#endif
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
      // The block that holds the bidder's previous best object, evaluated exactly: the lanes that hold its two
      // best values (more on ties) start the running top two; the second of them is a lower bound of the final
      // second-best value (64 distinct real objects).
      BidState st;
      st.b1 = -1e9f;
      st.b2 = -1e9f;
      st.bk = -1;
      st.b2k = -1;
      st.tm = __builtin_inff();
      {
        const float4 o = sh.obj[home * kResBlock + lane];
        const float v = emd_value(sqdist3(o.x - qx, o.y - qy, o.z - qz), o.w);
        float t1, t2;
        wave_top2(v, t1, t2);
        emd_fold(st, __ballot(v >= t2), v, home * kResBlock + lane, n, tpu, sc.perm);
      }
#ifdef MVP_EMD_PROFILE
      prof_seed += __builtin_readcyclecounter() - tb0;
#endif
      auto blocks = [&]() -> unsigned long long {
        const float tq = st.tm - bpl;
        return __ballot(tq >= 0.f && bd2 <= tq * tq);   // (lanes >= nblk: inf)
      };
      unsigned long long bm = blocks() & ~(1ull << home);
      while (bm) {
        // the next four surviving blocks (fewer: the last one again, its candidates masked out)
        int bi[4];
        bool ok[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ok[r] = bm != 0ull;
          bi[r] = ok[r] ? (int)__builtin_ctzll(bm) : bi[r > 0 ? r - 1 : 0];
          bm &= bm - 1ull;
        }
        float4 o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = sh.obj[bi[r] * kResBlock + lane];
        float sd[4];
        unsigned long long m[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sd[r] = sqdist3(o[r].x - qx, o[r].y - qy, o[r].z - qz);
          const float tq = st.tm - o[r].w;
          m[r] = ok[r] ? __ballot(tq >= 0.f && sd[r] <= tq * tq) : 0ull;
        }
#ifdef MVP_EMD_PROFILE
        prof_blocks += (int)ok[0] + (int)ok[1] + (int)ok[2] + (int)ok[3];
        prof_folds += __builtin_popcountll(m[0]) + __builtin_popcountll(m[1]) + __builtin_popcountll(m[2]) + __builtin_popcountll(m[3]);
#endif
        if ((m[0] | m[1] | m[2] | m[3]) != 0ull) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = emd_value(sd[r], o[r].w);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (m[r]) emd_fold(st, m[r], v[r], bi[r] * kResBlock + lane, n, tpu, sc.perm);
          if (bm) bm &= blocks();
        }
      }
#ifdef MVP_EMD_PROFILE
      prof_bidcyc += __builtin_readcyclecounter() - tb0;
      prof_nbid += 1;
#endif
      if (__builtin_expect(st.bk < 0 || st.b2k < 0, 0)) {   // cannot happen: a block holds 64 objects
        if (lane == 0) sh.s_err = 1;
        st.bk = st.bk < 0 ? 0 : st.bk;
      }
      if (lane == 0) {
        sh.s_bj[u] = j;
        sh.s_bo[u] = st.bk;
        sh.s_binc[u] = st.b1 - st.b2 + eps;
        sh.h1[j] = (unsigned short)st.bk;
        atomicAdd(&cnt[st.bk & (kResBuckets - 1)], 1);
      }
    }
    if (wave >= U && !last) {
      // a wave without a bidder refreshes a block's price bound (prices do not move during Bid: exact)
      const int blk = (int)(((unsigned)it * (unsigned)kEmdWaves + (unsigned)wave) % (unsigned)nblk);
      const float pm = wave_min_lane63(sh.obj[blk * kResBlock + lane].w);
      if (lane == kWave - 1) sh.b_lo[blk].w = pm;
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
    if (last) {
      // every bidder of the last round takes what it bid on (emd_cuda.cu:201-212): resolved below the loop
      last_u = U;
      break;
    }

    // ---------------- GetMax + Assign (emd_cuda.cu:181-215): every wave settles the bids it placed
    const int nxt = cur ^ 1;
    for (int u = wave; u < U; u += kEmdWaves) {
      const int j = __builtin_amdgcn_readfirstlane(sh.s_bj[u]), o = __builtin_amdgcn_readfirstlane(sh.s_bo[u]);
      const float inc = sh.s_binc[u];
      const int c = __builtin_amdgcn_readfirstlane(cnt[o & (kResBuckets - 1)]);
      int prev = sh.owner[o];
      const float price = sh.obj[o].w;
      bool win = true;
      if (__builtin_expect(c != 1, 0)) {
        // another bid in my bucket: compare the round's bids -- the maximal increment bid on my object, then
        // the highest bidder inside its 1e-6 band
#ifdef MVP_EMD_PROFILE
        prof_slow += 1;
#endif
        float mi = inc;
        for (int v = 0; v < U; ++v)
          if (sh.s_bo[v] == o) mi = __builtin_fmaxf(mi, sh.s_binc[v]);
        int wj = -1;
        for (int v = 0; v < U; ++v)
          if (sh.s_bo[v] == o && emd_in_band(sh.s_binc[v], mi)) wj = max(wj, sh.s_bj[v]);
        win = wj == j;
      }
      // next round's list: the evicted owner, or the loser itself (the order of a list changes no result)
      int again = j;
      if (win) {   // one winner per object
        again = prev == kResFree ? -1 : prev;
        if (lane == 0) {
          sh.owner[o] = (unsigned short)j;
          sh.obj[o].w = price + inc;
        }
      }
      if (again >= 0 && lane == 0) sh.list[nxt][atomicAdd(&sh.s_cnt[nxt], 1)] = again;   // (<= U entries: never grows)
    }
    if (wave == kEmdWaves - 1) {
      // the next round's counters (last read in the round before this one), the list size after next
      if (lane < kResBuckets / 4) reinterpret_cast<int4 *>(sh.cnt[(it + 1) & 1])[lane] = make_int4(0, 0, 0, 0);
      if (lane == 0) sh.s_cnt[cur] = 0;
    }
    lds_barrier();
    cur = nxt;
    U = __builtin_amdgcn_readfirstlane(sh.s_cnt[cur]);
#ifdef MVP_EMD_PROFILE
    cyc_assign += __builtin_readcyclecounter() - tp2;
#endif
  }
#ifdef MVP_EMD_PROFILE
  if (cloud < 2 && lane == 0 && (wave == 0 || wave == 3))
    printf("resident cloud %d wave %d: rounds %lld bids(all waves) %lld | this wave: %lld bids, %lld cycles each (home block %lld), blocks %.1f folds %.1f per bid | cycles bid %lld wait %lld assign %lld total %lld | contested buckets %lld\n",
           cloud, wave, n_rounds, n_bids, prof_nbid, prof_bidcyc / (prof_nbid + 1), prof_seed / (prof_nbid + 1), (double)prof_blocks / (double)(prof_nbid + 1),
           (double)prof_folds / (double)(prof_nbid + 1), cyc_bid, cyc_sync1, cyc_assign, __builtin_readcyclecounter() - t_loop0, prof_slow);
  if (cloud < 2 && lane == 0 && wave == 0)
    printf("resident cloud %d: %lld cycles in %lld ticks of the 100 MHz clock = %.0f MHz\n", cloud, __builtin_readcyclecounter() - t_loop0,
           wall_clock64() - w_loop0, 100.0 * (double)(__builtin_readcyclecounter() - t_loop0) / (double)(wall_clock64() - w_loop0));
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
