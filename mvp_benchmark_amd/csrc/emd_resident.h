// LDS-resident tail of the EMD auction for gfx950: clouds of at most 4096 points (device code; the kernels that
// run it are emd_resident_kernel in emd_resident.hip and -- fused, without a launch in between -- emd_lean_kernel).
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
#pragma once
#include "emd_common.h"

namespace mvp {

// (kResList, kResMaxN, kResMinRounds: emd_common.h)
constexpr int kResSub = 16;       // slots per sub-block: a 16-lane row of a visit step
#ifndef MVP_RES_SOLO
#define MVP_RES_SOLO 1            // the last bidder's chain of evictions on one wave, without barriers (0: A/B builds)
#endif
constexpr int kResBuckets = 256;
constexpr unsigned short kResFree = 0xFFFFu;
static_assert(kResList % kEmdWaves == 0, "list positions are dealt out to the waves round-robin, for the life of the launch");

template <int NMAX>
struct ResShared {
  float4 obj[NMAX];                      // slot -> x, y, z, price
  float px[NMAX], py[NMAX], pz[NMAX];    // person -> point
  unsigned short owner[NMAX];            // slot -> person (kResFree: none)
  unsigned short h1[NMAX];               // person -> slot it last bid on (seed hint); at the end: person -> slot
  float4 s_lo[NMAX / kResSub], s_hi[NMAX / kResSub];   // sub-block: box min + price lower bound / box max
  unsigned short w_list[kEmdWaves][NMAX / kResSub + 16];   // surviving sub-blocks of a wave's search (+ a step's over-read)
  int cnt[2][kResBuckets];               // by round parity: bids per bucket of object slots
  float4 r_q[kResList];                  // a position's bidder: its point, bits(person) (-1: nobody)
  int r_p1[kResList];                    // ... and the slot it last bid on (the home block of its next search)
  int s_bj[kResList], s_bo[kResList];    // this round's bid of every position (person or -1, slot)
  float s_binc[kResList];                // its increment
  int s_act[3];                          // by round % 3: positions that hold a bidder
  int s_err;
  int s_solo[2];                         // rounds the last bidder's wave ran alone; 1: it ran the forced last round
};

// min over each 16-lane row, valid in every lane of the row
__device__ __forceinline__ float row_min(float v) {
  const float inf = __builtin_inff();
  v = __builtin_fminf(v, dpp_f32<0xB1, 0xF>(inf, v));    // quad_perm [1,0,3,2]
  v = __builtin_fminf(v, dpp_f32<0x4E, 0xF>(inf, v));    // quad_perm [2,3,0,1]
  v = __builtin_fminf(v, dpp_f32<0x141, 0xF>(inf, v));   // row_half_mirror
  v = __builtin_fminf(v, dpp_f32<0x140, 0xF>(inf, v));   // row_mirror
  return v;
}

#ifdef MVP_EMD_PROFILE
#define RES_PROF_ARGS , long long &prof_seed, long long &prof_subs, long long &prof_folds
#define RES_PROF_PASS , prof_seed, prof_subs, prof_folds
#else
#define RES_PROF_ARGS
#define RES_PROF_PASS
#endif

// Bid of one person (emd_cuda.cu:95-179) by the calling wave against the LDS-resident state: the exact best / second-best
// value and the best slot, wave-uniform.  (qx, qy, qz): the person's point; p1: the slot it last bid on (its 64-slot block
// is evaluated first); wl: the wave's list of surviving sub-blocks.
template <int NMAX>
__device__ __forceinline__ BidState res_search(ResShared<NMAX> &sh, unsigned short *wl, const float qx, const float qy, const float qz,
                                               const int p1, const int lane, const int row, const int sl, const int n, const int nsub,
                                               const int npass, const int tpu, const int *__restrict__ perm RES_PROF_ARGS) {
  constexpr int kPasses = NMAX / kResSub / kWave;   // sub-block tests per lane this instantiation can need: 2 / 4
#ifdef MVP_EMD_PROFILE
  const long long tb0 = __builtin_readcyclecounter();
#endif
  const int home = p1 >> 6;   // the 64-slot block (four sub-blocks) that holds the previous best object
  // every sub-block's box and price bound against the bidder's point (independent of the seed: issued first)
  float bd2[kPasses], bpl[kPasses];
  // (straight-line: every pass the instantiation can need is loaded at once -- indices beyond this cloud's
  // sub-blocks are clamped and their result discarded -- so the loads share one LDS round trip)
#pragma unroll
  for (int ps = 0; ps < kPasses; ++ps) {
    const int sub = min(ps * kWave + lane, nsub - 1);
    const float4 lo = sh.s_lo[sub], hi = sh.s_hi[sub];
    const float dx = __builtin_fmaxf(__builtin_fmaxf(lo.x - qx, qx - hi.x), 0.f);
    const float dy = __builtin_fmaxf(__builtin_fmaxf(lo.y - qy, qy - hi.y), 0.f);
    const float dz = __builtin_fmaxf(__builtin_fmaxf(lo.z - qz, qz - hi.z), 0.f);
    bd2[ps] = ps < npass ? sqdist3(dx, dy, dz) : __builtin_inff();
    bpl[ps] = lo.w;
  }
  // The home block evaluated exactly: the lanes that hold its two best values (more on ties) start the
  // running top two; the second of them is a lower bound of the final second-best value (64 distinct objects).
  BidState st;
  st.b1 = -1e9f;
  st.b2 = -1e9f;
  st.bk = -1;
  st.b2k = -1;
  st.tm = __builtin_inff();
  {
    const float4 o = sh.obj[home * kWave + lane];
    const float v = emd_value(sqdist3(o.x - qx, o.y - qy, o.z - qz), o.w);
    // largest value, then the largest of the rest (one holder of the maximum set aside)
    const float t1 = emd_wave_max(v);
    const unsigned long long m1 = __ballot(v == t1);
    const int l1 = (int)__builtin_ctzll(m1);
    const float t2 = emd_wave_max(lane == l1 ? -1e9f : v);
    const unsigned long long m2 = __ballot(v >= t2);
    if (__builtin_expect(t1 > t2 && __builtin_popcountll(m2) == 2, 1)) {
      // two different values, one holder each: the state emd_fold would arrive at
      st.b1 = t1;
      st.bk = home * kWave + l1;
      st.b2 = t2;
      st.b2k = home * kWave + (int)__builtin_ctzll(m2 & ~m1);
      st.tm = (3.0f - t2) + kMargin;
    } else {
      emd_fold(st, m2, v, home * kWave + lane, n, tpu, perm);   // equal values: the reference's tie order
    }
  }
#ifdef MVP_EMD_PROFILE
  prof_seed += __builtin_readcyclecounter() - tb0;
#endif
  // surviving sub-blocks (the home block's four excluded) compacted into the wave's list
  int nl = 0;
#pragma unroll
  for (int ps = 0; ps < kPasses; ++ps) {
    const float tq = st.tm - bpl[ps];
    const int sub = ps * kWave + lane;
    const bool pass = tq >= 0.f && bd2[ps] <= tq * tq && (sub >> 2) != home;   // (passes beyond npass: inf)
    const unsigned long long m = __ballot(pass);
    if (pass) wl[nl + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (unsigned short)sub;
    nl += __builtin_popcountll(m);
  }
#ifdef MVP_EMD_PROFILE
  prof_subs += nl;
#endif
  // visit: a step = four sub-blocks, one per 16-lane row; four steps in flight
  for (int k0 = 0; k0 < nl; k0 += 16) {
    // (the list is read past its end -- the row is padded -- and the entry discarded: four independent reads)
    int ent[4], slot[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ent[r] = wl[k0 + 4 * r + row];
#pragma unroll
    for (int r = 0; r < 4; ++r) slot[r] = k0 + 4 * r + row < nl ? ent[r] * kResSub + sl : -1;
    float4 o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = sh.obj[slot[r] < 0 ? sl : slot[r]];
    float sd[4];
    unsigned long long m[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sd[r] = sqdist3(o[r].x - qx, o[r].y - qy, o[r].z - qz);
      const float tq = st.tm - o[r].w;
      m[r] = __ballot(slot[r] >= 0 && tq >= 0.f && sd[r] <= tq * tq);
    }
#ifdef MVP_EMD_PROFILE
    prof_folds += __builtin_popcountll(m[0]) + __builtin_popcountll(m[1]) + __builtin_popcountll(m[2]) + __builtin_popcountll(m[3]);
#endif
    // (exact values only for the steps that hold a candidate: 1-2 of the four, usually)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (m[r]) emd_fold(st, m[r], emd_value(sd[r], o[r].w), slot[r], n, tpu, perm);
  }
  return st;
}

// The remaining rounds of one cloud, on the calling workgroup (1024 threads), from the hand-over record and the
// lists the previous kernel -- or the calling kernel itself, a moment ago -- left in the scratch.
template <int NMAX>
__device__ __forceinline__ void emd_resident_body(ResShared<NMAX> &sh, const int cloud, int b, int n, float *__restrict__ dist,
                                                  int *assignment, float eps, int iters, char *scratch) {
  if (cloud >= b) return;
  const int t = threadIdx.x;
  const int lane = t & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // (tells the compiler that it is wave-uniform)
  const int row = lane >> 4, sl = lane & 15;
  char *tail = scratch + (size_t)b * emd_scratch_per_cloud(n);
  EmdHandover *resume = emd_handover(tail, b, cloud);
  long long *stats = emd_stats(tail, b, cloud);
  // (an L1-bypassing load: fused into the lean launch, the record was written a moment ago by this very workgroup)
  const int it0 = __hip_atomic_load(&resume->next_it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (it0 == 0) return;   // finished in an earlier launch (uniform over the workgroup)
  const EmdScratch sc = emd_carve(scratch + (size_t)cloud * emd_scratch_per_cloud(n), n);
  dist += (size_t)cloud * n;
  int *ass = assignment + (size_t)cloud * n;
  const int nsub = n / kResSub;          // n % 1024 == 0: a multiple of 64
  const int npass = nsub / kWave;        // ... and this cloud needs: 1, 2, 3 or 4

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
  // List positions: position p holds the p-th entry of the lists the previous launch left (its cluster width:
  // nlists), concatenated, and belongs to wave p % 16 for the rest of the auction: a loser stays at its
  // position, an evicted owner takes the place of the winner that evicted it, a winner of a free object leaves
  // the position empty.
  int npos = 0;   // positions in use at the hand-over (their number never grows)
  {
    const int nl = resume->nlists;
    int p = t, k = -1, total = 0;
#pragma unroll
    for (int w = 0; w < kMaxCluster; ++w) {
      const int cw = w < nl ? resume->cnt[w] : 0;
      if (k < 0 && p >= 0 && p < cw) k = sc.ulist[(size_t)w * 2 * n + p];
      p -= cw;
      total += cw;
    }
    npos = __builtin_amdgcn_readfirstlane(min(total, kResList));
    if (t < kResList) {
      const float4 pa = t < npos ? sc.person[2 * k] : make_float4(0.f, 0.f, 0.f, 0.f);
      const int p1 = t < npos ? __float_as_int(sc.person[2 * k + 1].y) : 0;
      sh.r_q[t] = make_float4(pa.x, pa.y, pa.z, __int_as_float(t < npos ? k : -1));
      sh.r_p1[t] = p1 < 0 ? 0 : p1;   // (every person has bid before a hand-over; any block would do)
    }
    if (t == 0) {
      sh.s_act[it0 % 3] = npos;
      sh.s_act[(it0 + 1) % 3] = 0;
      sh.s_act[(it0 + 2) % 3] = 0;
      sh.s_err = (resume->err != 0 || total > kResList) ? 1 : 0;   // (the launcher never hands over more)
      sh.s_solo[0] = 0;
      sh.s_solo[1] = 0;
    }
  }
  __syncthreads();
  // exact bounding box and exact price minimum per sub-block of 16 slots: a 16-lane row each
  for (int q = wave; q < nsub / 4; q += kEmdWaves) {
    const int sub = 4 * q + row;
    const float4 o = sh.obj[sub * kResSub + sl];
    float lx = o.x, ly = o.y, lz = o.z, hx = o.x, hy = o.y, hz = o.z;
#pragma unroll
    for (int off = 1; off < kResSub; off <<= 1) {
      lx = __builtin_fminf(lx, __shfl_xor(lx, off, kResSub));
      ly = __builtin_fminf(ly, __shfl_xor(ly, off, kResSub));
      lz = __builtin_fminf(lz, __shfl_xor(lz, off, kResSub));
      hx = __builtin_fmaxf(hx, __shfl_xor(hx, off, kResSub));
      hy = __builtin_fmaxf(hy, __shfl_xor(hy, off, kResSub));
      hz = __builtin_fmaxf(hz, __shfl_xor(hz, off, kResSub));
    }
    const float lw = row_min(o.w);
    if (sl == 0) {
      sh.s_lo[sub] = make_float4(lx, ly, lz, lw);
      sh.s_hi[sub] = make_float4(hx, hy, hz, 0.f);
    }
  }
  __syncthreads();

  // ------------------------------------------------------------ the auction
  long long n_rounds = 0, n_bids = 0;
  bool last_done = false;   // the forced last round ran: the positions' bids are their assignment
  // The single-bidder chain exists in the <= 2048-point instantiation only: a larger cloud is hardly ever down to one bidder
  // (mean 7-10 after the hand-over at 4096 points) and measured 2-3 % slower with the code present (NOTES_r6 §16).
  constexpr bool kSolo = MVP_RES_SOLO != 0 && NMAX <= 2048;
  [[maybe_unused]] int solo_it = -1;   // kSolo: the round from which a single bidder is left (uniform over the workgroup)
  unsigned short *wl = sh.w_list[wave];
#ifdef MVP_EMD_PROFILE
  long long prof_folds = 0, prof_subs = 0, prof_bidcyc = 0, prof_nbid = 0, cyc_bid = 0, cyc_sync1 = 0, cyc_assign = 0, prof_slow = 0,
            prof_seed = 0;
  const long long t_loop0 = __builtin_readcyclecounter();
  const long long w_loop0 = wall_clock64();
#endif
  for (int it = it0; it < iters; ++it) {
    const int U = __builtin_amdgcn_readfirstlane(sh.s_act[it % 3]);
    if (U == 0) break;
    n_rounds += 1;
    n_bids += U;
    const bool last = it == iters - 1;
    const int tpu = -U;   // thread_per_unass (emd_cuda.cu:107-109), resolved inside emd_precedes: ties only
    int *cnt = sh.cnt[it & 1];
    if ((U + kEmdWaves - 1) / kEmdWaves < (npos + kEmdWaves - 1) / kEmdWaves) {
      // Positions empty out at random: once the persons left would fit fewer positions per wave, wave 0 moves
      // them to the front (at most three times per cloud; the order of the list changes no result).
      if (wave == 0) {
        const float4 rq = sh.r_q[lane];
        const int rp = sh.r_p1[lane];
        const bool act = lane < npos && __float_as_int(rq.w) >= 0;
        const unsigned long long m = __ballot(act);
        const int dst = __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (act) {
          sh.r_q[dst] = rq;
          sh.r_p1[dst] = rp;
        }
        if (lane >= __builtin_popcountll(m) && lane < npos) sh.r_q[lane] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
      }
      lds_barrier();
      npos = U;
    }

    if constexpr (kSolo) {
      if (U == 1) {   // the number of bidders never grows: the rest of the auction is one chain of evictions (below the loop)
        solo_it = it;
        n_rounds -= 1;
        n_bids -= 1;
        break;
      }
    }
    // ---------------- Bid (emd_cuda.cu:95-179): the wave bids for the persons at its positions, one after the other
#ifdef MVP_EMD_PROFILE
    const long long tp0 = __builtin_readcyclecounter();
#endif
    bool idle = true;
    for (int pos = wave; pos < npos; pos += kEmdWaves) {
      const float4 rq = sh.r_q[pos];
      const int j = __builtin_amdgcn_readfirstlane(__float_as_int(rq.w));
      if (j < 0) {
        if (lane == 0) sh.s_bj[pos] = -1;
        continue;
      }
      idle = false;
      const float qx = rq.x, qy = rq.y, qz = rq.z;
      const int p1 = __builtin_amdgcn_readfirstlane(sh.r_p1[pos]);
#ifdef MVP_EMD_PROFILE
      const long long tb0 = __builtin_readcyclecounter();
#endif
      BidState st = res_search<NMAX>(sh, wl, qx, qy, qz, p1, lane, row, sl, n, nsub, npass, tpu, sc.perm RES_PROF_PASS);
#ifdef MVP_EMD_PROFILE
      prof_bidcyc += __builtin_readcyclecounter() - tb0;
      prof_nbid += 1;
#endif
      if (__builtin_expect(st.bk < 0 || st.b2k < 0, 0)) {   // cannot happen: a block holds 64 objects
        if (lane == 0) sh.s_err = 1;
        st.bk = st.bk < 0 ? 0 : st.bk;
      }
      if (lane == 0) {
        sh.s_bj[pos] = j;
        sh.s_bo[pos] = st.bk;
        sh.s_binc[pos] = st.b1 - st.b2 + eps;
        sh.r_p1[pos] = st.bk;   // the hint of this person's next bid (a loser bids again at once)
        sh.h1[j] = (unsigned short)st.bk;
        atomicAdd(&cnt[st.bk & (kResBuckets - 1)], 1);
      }
    }
    if (idle && !last) {
      // a wave without a bidder refreshes four sub-blocks' price bounds (prices do not move during Bid: exact)
      const int sub = 4 * (int)(((unsigned)it * (unsigned)kEmdWaves + (unsigned)wave) % (unsigned)(nsub / 4)) + row;
      const float pm = row_min(sh.obj[sub * kResSub + sl].w);
      if (sl == 0) sh.s_lo[sub].w = pm;
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
      last_done = true;
      break;
    }

    // ---------------- GetMax + Assign (emd_cuda.cu:181-215): every wave settles the bids it placed
    for (int pos = wave; pos < npos; pos += kEmdWaves) {
      int j = __builtin_amdgcn_readfirstlane(sh.s_bj[pos]);
      if (j < 0) continue;
      const int me = j;
      const int bk = __builtin_amdgcn_readfirstlane(sh.s_bo[pos]);
      const float inc = sh.s_binc[pos];
      const int c = __builtin_amdgcn_readfirstlane(cnt[bk & (kResBuckets - 1)]);
      const int prev = __builtin_amdgcn_readfirstlane((int)sh.owner[bk]);
      const float price = sh.obj[bk].w;
      bool win = true;
      if (__builtin_expect(c != 1, 0)) {
        // another bid in my bucket: compare the round's bids -- the maximal increment bid on my object, then
        // the highest bidder inside its 1e-6 band
#ifdef MVP_EMD_PROFILE
        prof_slow += 1;
#endif
        float mi = inc;
        for (int v = 0; v < npos; ++v)
          if (sh.s_bj[v] >= 0 && sh.s_bo[v] == bk) mi = __builtin_fmaxf(mi, sh.s_binc[v]);
        int wj = -1;
        for (int v = 0; v < npos; ++v)
          if (sh.s_bj[v] >= 0 && sh.s_bo[v] == bk && emd_in_band(sh.s_binc[v], mi)) wj = max(wj, sh.s_bj[v]);
        win = wj == j;
      }
      if (win) {   // one winner per object; the evicted owner takes this position
        j = prev == kResFree ? -1 : prev;
        const int jc = j < 0 ? 0 : j;
        const float nx = sh.px[jc], ny = sh.py[jc], nz = sh.pz[jc];
        const int np1 = sh.h1[jc];
        if (lane == 0) {
          sh.owner[bk] = (unsigned short)me;
          sh.obj[bk].w = price + inc;
          sh.r_q[pos] = make_float4(nx, ny, nz, __int_as_float(j));
          sh.r_p1[pos] = np1 == kResFree ? 0 : np1;
        }
      }
      if (j >= 0 && lane == 0) atomicAdd(&sh.s_act[(it + 1) % 3], 1);
    }
    if (wave == kEmdWaves - 1) {
      // the next round's counters (last read in the round before this one), the count of the round after next
      if (lane < kResBuckets / 4) reinterpret_cast<int4 *>(sh.cnt[(it + 1) & 1])[lane] = make_int4(0, 0, 0, 0);
      if (lane == 0) sh.s_act[(it + 2) % 3] = 0;
    }
    lds_barrier();
#ifdef MVP_EMD_PROFILE
    cyc_assign += __builtin_readcyclecounter() - tp2;
#endif
  }
#ifdef MVP_EMD_PROFILE
  if (cloud < 2 && lane == 0 && (wave == 0 || wave == 3))
    printf("resident cloud %d wave %d: rounds %lld bids(all waves) %lld | this wave: %lld bids, %lld cycles each (home block %lld), sub-blocks %.1f folds %.1f per bid | cycles bid %lld wait %lld assign %lld total %lld | contested buckets %lld\n",
           cloud, wave, n_rounds, n_bids, prof_nbid, prof_bidcyc / (prof_nbid + 1), prof_seed / (prof_nbid + 1), (double)prof_subs / (double)(prof_nbid + 1),
           (double)prof_folds / (double)(prof_nbid + 1), cyc_bid, cyc_sync1, cyc_assign, __builtin_readcyclecounter() - t_loop0, prof_slow);
  if (cloud < 2 && lane == 0 && wave == 0)
    printf("resident cloud %d: %lld cycles in %lld ticks of the 100 MHz clock = %.0f MHz\n", cloud, __builtin_readcyclecounter() - t_loop0,
           wall_clock64() - w_loop0, 100.0 * (double)(__builtin_readcyclecounter() - t_loop0) / (double)(wall_clock64() - w_loop0));
#endif

  // ------------------------------------------------------------ one bidder left: a chain of evictions on ONE wave
  // A single bid is never contested and the person it evicts is the next round's only bidder: the wave that holds the
  // position runs Bid and Assign of every remaining round back to back -- no barrier, no counters, the next bidder's
  // point and hint straight from the owner it evicted -- and the other fifteen wait at the barrier below.  Same bids,
  // same increments, same rounds as the two-barrier round above (20-37 % of the rounds of a 1024-point cloud).
  if constexpr (kSolo) if (solo_it >= 0) {
    int mypos = -1;
    for (int pos = wave; pos < npos; pos += kEmdWaves) {
      if (__builtin_amdgcn_readfirstlane(__float_as_int(sh.r_q[pos].w)) >= 0)
        mypos = pos;
      else if (lane == 0)
        sh.s_bj[pos] = -1;   // (a bid of an earlier round must not be taken for one of the forced last round)
    }
    if (mypos >= 0) {
      const float4 rq = sh.r_q[mypos];
      int j = __builtin_amdgcn_readfirstlane(__float_as_int(rq.w));
      float qx = rq.x, qy = rq.y, qz = rq.z;
      int p1 = __builtin_amdgcn_readfirstlane(sh.r_p1[mypos]);
      int ran = 0, forced = 0;
      for (int it = solo_it; it < iters; ++it) {
        ran += 1;
        BidState st = res_search<NMAX>(sh, wl, qx, qy, qz, p1, lane, row, sl, n, nsub, npass, -1, sc.perm RES_PROF_PASS);
        if (__builtin_expect(st.bk < 0 || st.b2k < 0, 0)) {   // cannot happen: a block holds 64 objects
          if (lane == 0) sh.s_err = 1;
          st.bk = st.bk < 0 ? 0 : st.bk;
        }
        const int bk = st.bk;
        if (it == iters - 1) {   // the forced last round (emd_cuda.cu:201-212): the bidder takes what it bid on
          if (lane == 0) {
            sh.s_bj[mypos] = j;
            sh.s_bo[mypos] = bk;
          }
          forced = 1;
          break;
        }
        // Assign: the bidder wins; the sub-block's price bound follows the one price that moved
        const int prev = __builtin_amdgcn_readfirstlane((int)sh.owner[bk]);
        const float pw = sh.obj[(bk & ~(kResSub - 1)) + sl].w;
        const float np = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pw), bk & (kResSub - 1))) + (st.b1 - st.b2 + eps);
        const float pm = row_min(sl == (bk & (kResSub - 1)) ? np : pw);
        if (lane == 0) {
          sh.owner[bk] = (unsigned short)j;
          sh.h1[j] = (unsigned short)bk;   // the hint of its next bid, should it be evicted again
          sh.obj[bk].w = np;
          sh.s_lo[bk / kResSub].w = pm;
        }
        if (prev == kResFree) break;   // a free object: everybody is assigned
        j = prev;
        qx = sh.px[prev];
        qy = sh.py[prev];
        qz = sh.pz[prev];
        const int np1 = __builtin_amdgcn_readfirstlane((int)sh.h1[prev]);
        p1 = np1 == kResFree ? 0 : np1;
      }
      if (lane == 0) {
        sh.s_solo[0] = ran;
        sh.s_solo[1] = forced;
      }
    }
  }

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
  if constexpr (kSolo) {
    n_rounds += sh.s_solo[0];
    n_bids += sh.s_solo[0];
    last_done = last_done || sh.s_solo[1] != 0;
  }
  if (last_done && t < npos && sh.s_bj[t] >= 0) pslot[sh.s_bj[t]] = (unsigned short)sh.s_bo[t];
  __syncthreads();
  for (int p = t; p < n; p += kEmdThreads) {
    const int s = pslot[p];
    const float4 o = sh.obj[s];
    dist[p] = sqdist3(sh.px[p] - o.x, sh.py[p] - o.y, sh.pz[p] - o.z);
    ass[p] = sc.perm[s];
  }
  if (t == 0) {
    atomicAdd(reinterpret_cast<unsigned long long *>(&stats[0]), (unsigned long long)n_rounds);
    atomicAdd(reinterpret_cast<unsigned long long *>(&stats[1]), (unsigned long long)n_bids);
    if (sh.s_err) atomicAdd(reinterpret_cast<unsigned long long *>(&stats[0]), (unsigned long long)(-(1ll << 40)));
    resume->next_it = 0;              // finished
    resume->last_width = 1 + 16 * 3;  // one workgroup, the resident launch
  }
}

}  // namespace mvp
