// The tail of the EMD auction: one workgroup per cloud, prices in LDS.
//
// Same reference semantics as emd.hip (emd_cuda.cu:95-215: Bid, GetMax, Assign
// per round; results bit-identical to the exhaustive oracle), for the rounds
// after the clustered kernel handed the cloud over: at most kTailCap = 256
// persons are still unassigned (their number never grows: a winner on a free
// object lowers it, a winner on an owned object replaces one unassigned person
// by another).  At the headline shape (64 clouds of 16384 points, eps 0.004,
// 3000 rounds) that is round ~150 onwards: 95 % of the rounds, 60 % of the
// bids, ~57 bidders per round -- a latency chain, not a throughput problem.
// In the clustered kernel such a round costs ~40k cycles: every bid is two
// dependent L2 round trips on state other CUs write, and a round is closed by
// two cross-CU all-gathers.  Here:
//   * the PRICES (the only state a bid reads besides the static coordinates)
//     live in LDS (n <= 16384 floats), together with the cell index: per cell
//     one 16-byte record {fp16 bounding box rounded outwards, price lower
//     bound} -- one ds_read_b128 per cell test.  A bid touches no state that
//     another CU writes;
//   * W workgroups (1, 2, 4 or 8, as in the clustered kernel) share a cloud
//     WITHOUT sharing that state: each keeps its own full copy of the prices
//     and cell bounds in its LDS and bids for its own part of the unassigned
//     persons; the round's bids (16 bytes each, <= 256 per cloud) are exchanged
//     through memory as self-validating records -- two 8-byte atomic stores,
//     each carrying the round's tag, polled by the readers: the data is the
//     flag, there is no barrier granule, no store drain and no all-gather in a
//     round -- and every workgroup then resolves all of them and applies all
//     price rises to its own copy: redundant, deterministic, identical.  (Every
//     member also derives every member's bidder count from the bids it has
//     seen, so it knows where to look.);
//   * the unassigned persons sit in a pool of kTailCap LDS entries {point,
//     hints, candidate cache}.  An entry is stable: a loser keeps it, a winner
//     hands it to the person it evicted (at most one), so a member's pool never
//     grows and needs no allocation; an order list of the live entries is
//     rebuilt by ballot each round.  With W > 1 a pool holds <= 128 entries and
//     the OWNER of every object is replicated in LDS too (u16), so finding the
//     evicted person is an LDS read instead of a cross-CU round trip;
//   * exact CANDIDATE CACHE per person.  Values v_k = 3 - d_k - price_k only
//     fall (prices only rise).  A full search keeps its filter `delta` looser
//     than the second-best value needs and so sees EVERY object within delta
//     of it; the (<= 16) best of them are cached as {slot, sqrtf distance}
//     with a bound tau >= the value of every object that is not cached.  A
//     later bid of that person re-evaluates the cached candidates from LDS
//     prices (bit-identical values: same sqrtf result, same double expression)
//     and is exact whenever their best is > tau and their second best >= tau
//     -- true for ~2/3 of the tail's bids, which then cost a few hundred
//     cycles and no memory round trip; otherwise the cached second best seeds
//     the full search (no seeding round trip) and the cache is rebuilt.  Caches
//     follow their person out of the pool (scratch, 128 B each) and back in;
//   * GetMax is resolved exactly in LDS (emd_cuda.cu:181-194: the highest bidder
//     inside the 1e-6 band of the object's maximal increment).  The bids of one
//     object find each other through the object's owner word: the last of them
//     to write its number there represents the group, count and maximal
//     increment are two LDS atomics on the representative -- O(1) per bid, no
//     alarm pass (W == 1 has no owner words in LDS and scans the round's bids).
// Bid phase A: every bidder's cache is tried, four bidders per wave (a 16-lane
// row each: 16 cached candidates).  Phase B: the misses, one wave each, drawn
// from a list so that the expensive searches spread over the 16 waves: cell
// enumeration from LDS, then ONE global round trip for the coordinates of the
// surviving cells' members (static data, plain cached loads), prices from LDS;
// every lane keeps the exact top two of the objects it evaluated, the wave
// merges them once (DPP, reference tie order on original indices).
// Inside a round the phases are separated by bare s_barrier (LDS traffic only),
// except one __syncthreads after phase A that also waits for the previous
// round's stores (records and caches of persons that left the pool) before the
// first bid record of the round is published.
#include "emd_common.h"

namespace mvp {

template <int W>
__global__ __launch_bounds__(kEmdThreads) void emd_tail_kernel(
    int b, int bpad, int n, const float *__restrict__ xyz1, float *__restrict__ dist, int *assignment, float eps,
    int iters, char *scratch, float delta) {
  constexpr int POOL = W == 1 ? kTailCap : kTailCap / 2;  // entries a member can hold (they never grow)
  constexpr bool LOWN = W > 1;                             // owners replicated in LDS
  // block -> (cloud, member) as in the clustered kernel: members bpad blocks apart share an XCD
  const int cloud = W == 1 ? (int)blockIdx.x : (int)blockIdx.x % bpad;
  const int wg = W == 1 ? 0 : (int)blockIdx.x / bpad;
  if (cloud >= b) return;
  const size_t per_cloud = emd_scratch_per_cloud(n);
  char *cbase = scratch + (size_t)cloud * per_cloud;
  char *tail = scratch + (size_t)b * per_cloud;
  EmdResume *resume = emd_resume(tail, b, cloud);
  long long *stats = emd_stats(tail, b, cloud);
  u64 *slots = emd_granules(tail, b, cloud, 1);
  const int it0 = resume->next_it;
  if (it0 == 0) return;  // the cloud was finished by the first kernel (uniform over the cluster)

  const int t = threadIdx.x;
  const int lane = t & (kWave - 1);
  const int wave = t >> 6;
  const int row = lane >> 4, l16 = lane & 15, rsh = row * 16;
  xyz1 += (size_t)cloud * n * 3;
  dist += (size_t)cloud * n;
  int *ass = assignment + (size_t)cloud * n;
  const EmdScratch sc = emd_carve(cbase, n);

  // Mutable global state (assignment, person records, caches, exchanged bids; owners when
  // W == 1) is read with L1-bypassing loads and written through.
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(cbase, 0, (int)per_cloud, 0x00020000);
  const unsigned off_person = (unsigned)n * 32u;
  const unsigned off_cache = (unsigned)(reinterpret_cast<char *>(sc.cache) - cbase);
  auto ldg16 = [&](unsigned byte_off) -> v4u { return __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16); };
  auto stg16 = [&](v4u v, unsigned byte_off) { __builtin_amdgcn_raw_buffer_store_b128(v, rs, byte_off, 0, 16); };
  auto ld_i32 = [&](int *p) -> int { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto st_i32 = [&](int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  // The round's bids travel through the (dead) unassigned lists of the first kernel, per member
  // and round parity: two 8-byte words {object | person << 16, tag} and {bits(increment),
  // (second-best hint + 1) | tag << 16}, tag = round + 1.
  u64 *xb = reinterpret_cast<u64 *>(cbase + (size_t)n * 68);
  auto xb_at = [&](int parity, int w, int i) -> u64 * { return xb + (size_t)((parity * kMaxCluster + w) * kTailCap + i) * 2; };

  __shared__ float s_price[kTailMaxN];
  __shared__ unsigned short s_owner[LOWN ? kTailMaxN : 2];       // 0xFFFF: free
  __shared__ unsigned s_owned[LOWN ? 2 : kTailMaxN / 32];        // W == 1: bit per object "has an owner"
  __shared__ uint4 s_cell[kTailCells + 1];  // {half2 x lo|hi, half2 y, half2 z, bits(price lower bound)}
  __shared__ unsigned short c_start[kTailCells + 3];
  // the pool of unassigned persons
  __shared__ float4 e_q[POOL];                              // the person's point
  __shared__ int4 e_i[POOL];                                // {person, hint slot 1, hint slot 2, cached candidates}
  __shared__ __attribute__((aligned(16))) unsigned short e_cs[POOL][kTailK];  // cached slots
  __shared__ __attribute__((aligned(16))) float e_cd[POOL][kTailK];           // their sqrtf distances
  __shared__ float e_tau[POOL];                             // >= value of every object NOT cached
  __shared__ unsigned char e_live[POOL];
  __shared__ unsigned short s_order[POOL];                  // live entries, ascending
  __shared__ unsigned short s_miss[POOL];                   // list positions whose cache failed
  __shared__ float s_seed[POOL];                            // ... and the seed their cache gives
  __shared__ unsigned short s_be[POOL];                     // own bids: the bidder's entry
  // this round's bids: first this member's own (phase A), then the whole cloud's
  __shared__ __attribute__((aligned(16))) int s_bo[kTailCap];
  __shared__ int s_b2k[kTailCap], s_bj[kTailCap];
  __shared__ __attribute__((aligned(16))) float s_binc[kTailCap];
  __shared__ unsigned char s_flag[kTailCap];                // phase A: own bid u was a cache hit
  __shared__ unsigned char s_bm[kTailCap];                  // member a bid came from
  __shared__ int s_gcnt[LOWN ? kTailCap : 2];               // bids on the object a bid represents
  __shared__ unsigned s_gmax[LOWN ? kTailCap : 2];          // ... and their maximal increment (ordered bits)
  __shared__ unsigned short w_list[kEmdWaves][128];         // surviving cells of a search
  __shared__ int st_slot[kEmdWaves][kStage];                // staged candidates of a search
  __shared__ float st_v[kEmdWaves][kStage], st_d[kEmdWaves][kStage];
  __shared__ int s_mcnt[kMaxCluster];                       // bidders per member (tracked identically by everybody)
  __shared__ int s_next, s_nmiss, s_err, s_U, s_nfree, s_abort;
  __shared__ unsigned s_gout[2 * kMaxCluster];
#ifdef MVP_EMD_PROFILE
  // [0] hits [1] misses [3] cycles in misses [4] linear scans [5] sum nsub [6] sum cells visited [8] home-cell seeds [11] visit cycles
  __shared__ unsigned long long s_prof[16];
  if (threadIdx.x < 16) s_prof[threadIdx.x] = 0ull;
  long long cyc_a = 0, cyc_b = 0, cyc_fetch = 0, cyc_resolve = 0, cyc_assign = 0, cyc_compact = 0,
            cyc_setup = __builtin_readcyclecounter();
#endif

  GridGeom gg;
  gg.g = resume->g;
  gg.lox = resume->lox;
  gg.loy = resume->loy;
  gg.loz = resume->loz;
  gg.invh = resume->invh;
  const int ncell = gg.g * gg.g * gg.g;
  const int U0 = resume->utot;
  if (ncell > kTailCells || U0 > kTailCap || U0 <= 0) {  // cannot happen for n <= kTailMaxN
    if (t == 0) stats[0] = -1;
    return;
  }

  // ------------------------------------------------------------ load the state
  for (int c = t; c <= ncell; c += kEmdThreads) c_start[c] = (unsigned short)sc.cstart[c];
  for (int s = t; s < n; s += kEmdThreads) {  // n % 1024 == 0: whole waves
    s_price[s] = sc.obj[s].w;
    const int ow = sc.ostate[s].z;
    if constexpr (LOWN) {
      s_owner[s] = (unsigned short)(ow < 0 ? 0xFFFF : ow);
    } else {
      const unsigned long long om = __ballot(ow != -1);
      if (lane == 0) {
        s_owned[(s >> 6) * 2] = (unsigned)om;
        s_owned[(s >> 6) * 2 + 1] = (unsigned)(om >> 32);
      }
    }
  }
  // member wg takes the list positions wg, wg + W, ...
  const int myU0 = (U0 - wg + W - 1) / W;
  if (t < POOL) {
    const bool live = t < myU0;
    e_live[t] = live ? 1 : 0;
    s_order[t] = (unsigned short)t;
    if (live) {
      const int j = resume->list[t * W + wg];
      const float4 lo = sc.person[2 * j];
      const float4 hi = sc.person[2 * j + 1];
      e_q[t] = lo;
      e_i[t] = make_int4(j, __float_as_int(hi.y), __float_as_int(hi.z), 0);  // caches start empty
    }
  }
  if (t < W) s_mcnt[t] = (U0 - t + W - 1) / W;
  if (t == 0) {
    s_next = kEmdWaves;
    s_nmiss = 0;
    s_err = resume->pad;
    s_U = myU0;
    s_nfree = 0;
    s_abort = 0;
  }
  unsigned epoch = 0;
  if constexpr (W > 1) {
    // no stale word of the exchange area may look like a record: clear this member's slices,
    // then meet the others once
    for (int i = t; i < 2 * kTailCap; i += kEmdThreads) {
      u64 *p = xb_at(i / kTailCap, wg, i % kTailCap);
      __hip_atomic_store(p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!emd_cluster_gather<W>(slots, wg, ++epoch, &s_U, &s_err, s_gout, &s_abort)) {
      if (t == 0) stats[0] = -2;
      for (int j = t; j < n; j += kEmdThreads) {
        dist[j] = __builtin_nanf("");
        ass[j] = -1;
      }
      return;
    }
  } else {
    __syncthreads();
  }
  // cell records: exact box of the members, rounded outwards to fp16; cheapest member
  for (int c = t; c < ncell; c += kEmdThreads) {
    float bx0 = __builtin_inff(), by0 = __builtin_inff(), bz0 = __builtin_inff();
    float bx1 = -__builtin_inff(), by1 = -__builtin_inff(), bz1 = -__builtin_inff();
    float pm = __builtin_inff();
    for (int s = c_start[c]; s < c_start[c + 1]; ++s) {
      const float4 o = sc.obj[s];
      bx0 = __builtin_fminf(bx0, o.x);
      by0 = __builtin_fminf(by0, o.y);
      bz0 = __builtin_fminf(bz0, o.z);
      bx1 = __builtin_fmaxf(bx1, o.x);
      by1 = __builtin_fmaxf(by1, o.y);
      bz1 = __builtin_fmaxf(bz1, o.z);
      pm = __builtin_fminf(pm, o.w);
    }
    uint4 r;
    r.x = half_bits_down(bx0) | (half_bits_up(bx1) << 16);
    r.y = half_bits_down(by0) | (half_bits_up(by1) << 16);
    r.z = half_bits_down(bz0) | (half_bits_up(bz1) << 16);
    r.w = __float_as_uint(c_start[c] < c_start[c + 1] ? pm : 0.f);
    s_cell[c] = r;
  }
  __syncthreads();

#ifdef MVP_EMD_PROFILE
  cyc_setup = __builtin_readcyclecounter() - cyc_setup;
#endif
  // ------------------------------------------------------------ the rounds
  const int block_cnt = n / 1024;
  const bool caching = delta > 0.f;
  long long n_rounds = 0, n_bids = 0;
  int U = myU0;      // this member's bidders
  int Utot = U0;     // the cloud's
  bool aborted = false;
  for (int it = it0; it < iters; ++it) {
    if (Utot == 0) break;
    n_rounds += 1;
    n_bids += U;
    const bool last = it == iters - 1;
    const unsigned tag = (unsigned)(it + 1);
    // thread_per_unass of the reference (emd_cuda.cu:107-109): fixes the tie order only
    const int upb = (Utot + block_cnt - 1) / block_cnt;
    const int tpu = 1024 / upb;
    // publish a bid of this member (list position u) to the others
    auto publish = [&](int u, int bk, float inc, int j, int b2k) {
      u64 *p = xb_at(it & 1, wg, u);
      __hip_atomic_store(p, ((u64)tag << 32) | (u64)((unsigned)bk | ((unsigned)j << 16)), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p + 1,
                         ((u64)(((tag & 0xFFFFu) << 16) | ((unsigned)(b2k + 1) & 0xFFFFu)) << 32) | (u64)__float_as_uint(inc),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

#ifdef MVP_EMD_PROFILE
    const long long tp0 = __builtin_readcyclecounter();
#endif
    // ---------------- Bid, phase A: the cached candidates at today's prices, four bidders per
    // wave (one per 16-lane row).  Exact when the best is > tau and the second best >= tau
    // (every object outside the cache is worth <= tau); a tie for the best goes to the search.
    for (int ub = 0; ub < U; ub += 4 * kEmdWaves) {
      const int u = ub + wave * 4 + row;
      const bool act = u < U;
      int e = 0, cc = 0, j = -1;
      if (act) {
        e = s_order[u];
        const int4 rb = e_i[e];
        j = rb.x;
        cc = rb.w;
      }
      float v = -1e9f;
      int slot = -1;
      const bool mine = act && l16 < cc;
      if (mine) {
        slot = e_cs[e][l16];
        v = emd_value_d(e_cd[e][l16], s_price[slot]);
      }
      float c1 = v, c2 = -1e9f;
      top2_dpp_step<0xB1, 0xF>(c1, c2);   // butterfly inside the row: every lane ends with
      top2_dpp_step<0x4E, 0xF>(c1, c2);   // the row's (largest, second largest with multiplicity)
      top2_dpp_step<0x141, 0xF>(c1, c2);
      top2_dpp_step<0x140, 0xF>(c1, c2);
      const float tau = act ? e_tau[e] : 0.f;
      const unsigned rm1 = (unsigned)((__ballot(mine && v == c1) >> rsh) & 0xFFFFull);
      const unsigned rm2a = (unsigned)((__ballot(mine && v == c2) >> rsh) & 0xFFFFull);
      const bool hit = act && cc >= 2 && c2 >= tau && c1 > tau && __builtin_popcount(rm1) == 1;
      const int l1 = rm1 ? __builtin_ctz(rm1) : 0;
      const unsigned rm2 = rm2a & ~(1u << l1);
      const int l2 = rm2 ? __builtin_ctz(rm2) : l1;
      const int bk = __shfl(slot, rsh + l1, kWave);
      const int k2 = __shfl(slot, rsh + l2, kWave);
      if (act && l16 == 0) {
        s_be[u] = (unsigned short)e;
        s_flag[u] = hit ? 1 : 0;
        if (hit) {
          s_bo[u] = bk;
          s_binc[u] = c1 - c2 + eps;
          s_bj[u] = j;
          s_b2k[u] = rm2 ? k2 : -1;
        } else {
          const int pos = atomicAdd(&s_nmiss, 1);
          s_miss[pos] = (unsigned short)u;
          s_seed[u] = cc >= 2 ? c2 : -1e9f;  // two distinct real objects reach it: bounds the final second best
        }
      }
    }
    // W > 1: everything this member stored in the previous round (records and caches of the
    // persons that left its pool) has landed before its first bid of this round is visible
    if constexpr (W > 1) __syncthreads();
    else lds_barrier();
    if constexpr (W > 1) {
      if (t < U && s_flag[t]) publish(t, s_bo[t], s_binc[t], s_bj[t], s_b2k[t]);
      // heartbeat (record kTailCap - 1; a pool holds <= kTailCap / 2 bids): also a member without
      // bidders tells the others that its stores of the previous round have landed
      if (t == kEmdThreads - 1)
        __hip_atomic_store(xb_at(it & 1, wg, kTailCap - 1), (u64)tag << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef MVP_EMD_PROFILE
    const long long tpa = __builtin_readcyclecounter();
#endif

    // ---------------- Bid, phase B (emd_cuda.cu:95-179): the misses, one wave each, drawn from
    // a counter so that the searches spread over the waves
    const int nmiss = s_nmiss;
    int mi = wave;
    for (int guard = 0; guard <= POOL && mi < nmiss; ++guard) {
      const int u = s_miss[mi];
      const int e = s_be[u];
      const float4 ra = e_q[e];
      const int4 rb = e_i[e];
      const int j = rb.x, p1 = rb.y, p2 = rb.z;
      const float qx = ra.x, qy = ra.y, qz = ra.z;
      float seed_b2 = s_seed[u];
#ifdef MVP_EMD_PROFILE
      const long long tb0 = __builtin_readcyclecounter();
      long long tvis = 0;
      int pcells = 0, plin = 0, pnsub = 0, phome = 0;
#endif
      // (1) seed: without a usable cache, the second-largest exact value among the home
      // cell's members and the previous best / second best
      if (seed_b2 == -1e9f) {
#ifdef MVP_EMD_PROFILE
        phome = 1;
#endif
        const int c0 = emd_cell(gg, qx, qy, qz);
        float a1 = -1e9f, a2 = -1e9f;
        const int s0 = c_start[c0], s1 = c_start[c0 + 1];
        for (int s = s0 + lane; s < s1; s += kWave) {
          const float4 o = sc.obj[s];
          top2_insert(a1, a2, emd_value(sqdist3(o.x - qx, o.y - qy, o.z - qz), s_price[s]));
        }
        bool extra = false;
        if ((lane == 0 && p1 >= 0) || (lane == 1 && p2 >= 0)) {
          const int ps = lane == 0 ? p1 : p2;
          const float4 o = sc.obj[ps];
          if (emd_cell(gg, o.x, o.y, o.z) != c0) {
            extra = true;
            top2_insert(a1, a2, emd_value(sqdist3(o.x - qx, o.y - qy, o.z - qz), s_price[ps]));
          }
        }
        const int have = (s1 - s0) + __builtin_popcountll(__ballot(extra));
        if (have < 2) {  // wave-uniform; rare: the first 64 slots (distinct objects)
          const float4 o = sc.obj[lane];
          a1 = emd_value(sqdist3(o.x - qx, o.y - qy, o.z - qz), s_price[lane]);
          a2 = -1e9f;
        }
        seed_b2 = wave_second_largest(a1, a2);
      }
      // The filter admits everything within delta of the seed (emd_common.h, kMargin, with
      // B2 := fl(seed - delta)): what it skips is worth < seed - delta <= (final second best) - delta.
      // It is not tightened during the search: a bid visits its surviving cells in one or two steps.
      const float tm = (3.0f - (seed_b2 - delta)) + kMargin;
      int nst = 0;                              // staged candidates (wave-uniform)
      float lb1 = -1e9f, lb2 = -1e9f;           // exact top two of the objects THIS LANE evaluates
      int lbk = -1, lb2k = -1;

      // evaluate one object per lane: filter, stage for the cache, lane-local top two
      auto consider = [&](bool valid, int s, const float4 &o) {
        const float p = valid ? s_price[s] : 0.f;
        const float sd = sqdist3(o.x - qx, o.y - qy, o.z - qz);
        const float tq = tm - p;
        const bool ps = valid && tq >= 0.f && sd <= tq * tq;
        const unsigned long long m = __ballot(ps);
        if (m) {
          const float d = __builtin_sqrtf(sd);
          const float v = emd_value_d(d, p);
          if (caching) {
            const int pos = nst + __builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (ps && pos < kStage) {
              st_slot[wave][pos] = s;
              st_v[wave][pos] = v;
              st_d[wave][pos] = d;
            }
            nst += __builtin_popcountll(m);
          }
          if (ps) {
            if (v > lb1) {
              lb2 = lb1; lb2k = lbk; lb1 = v; lbk = s;
            } else if (v == lb1) {   // rare: reference order on ORIGINAL indices
              lb2 = v;
              if (emd_precedes(sc.perm[s], sc.perm[lbk], n, tpu)) {
                lb2k = lbk; lbk = s;
              } else {
                lb2k = s;
              }
            } else if (v > lb2) {
              lb2 = v; lb2k = s;
            }
          }
        }
      };

      // (2) cells intersecting the cube |o - q|_inf <= tm (prices >= 0), 64 per step
      int ix0, iy0, iz0, nx, ny, nz;
      {
        const float r = tm * gg.invh + 1e-3f;  // slack covers index rounding
        const float fx = (qx - gg.lox) * gg.invh;
        const float fy = (qy - gg.loy) * gg.invh;
        const float fz = (qz - gg.loz) * gg.invh;
        const float gm = (float)(gg.g - 1);
        ix0 = (int)__builtin_fminf(__builtin_fmaxf(__builtin_floorf(fx - r), 0.f), gm);
        iy0 = (int)__builtin_fminf(__builtin_fmaxf(__builtin_floorf(fy - r), 0.f), gm);
        iz0 = (int)__builtin_fminf(__builtin_fmaxf(__builtin_floorf(fz - r), 0.f), gm);
        nx = (int)__builtin_fminf(__builtin_fmaxf(__builtin_floorf(fx + r), 0.f), gm) - ix0 + 1;
        ny = (int)__builtin_fminf(__builtin_fmaxf(__builtin_floorf(fy + r), 0.f), gm) - iy0 + 1;
        nz = (int)__builtin_fminf(__builtin_fmaxf(__builtin_floorf(fz + r), 0.f), gm) - iz0 + 1;
        ix0 = __builtin_amdgcn_readfirstlane(ix0);
        iy0 = __builtin_amdgcn_readfirstlane(iy0);
        iz0 = __builtin_amdgcn_readfirstlane(iz0);
        nx = __builtin_amdgcn_readfirstlane(nx);
        ny = __builtin_amdgcn_readfirstlane(ny);
        nz = __builtin_amdgcn_readfirstlane(nz);
      }
      const int nxy = nx * ny;
      const int nsub = nxy * nz;
      const float inv_nxy = __builtin_amdgcn_rcpf((float)nxy), inv_nx = __builtin_amdgcn_rcpf((float)nx);
      unsigned short *wl = w_list[wave];
      int nlist = 0;
      // a search cube covering most of the grid: scan the cell-sorted objects linearly
      const bool linear = 2 * nsub > ncell;
#ifdef MVP_EMD_PROFILE
      plin = linear ? 1 : 0;
      pnsub = nsub;
#endif
      if (linear) {
        for (int base = 0; base < n; base += kWave) {  // n % 1024 == 0
          const float4 o = sc.obj[base + lane];
          consider(true, base + lane, o);
        }
      }
      int cb = 0;
      bool enum_done = linear;
      for (;;) {
        // enumerate until the list holds at least 64 cells or the sub-box is exhausted
        while (!enum_done && nlist <= 128 - kWave) {
          const int i = cb + lane;
          bool cpass = false;
          int c = 0;
          if (i < nsub) {
            // exact small-integer division via float (i < 1331, divisors <= 121)
            const int kz = (int)(((float)i + 0.5f) * inv_nxy);
            const int rem = i - kz * nxy;
            const int ky = (int)(((float)rem + 0.5f) * inv_nx);
            const int kx = rem - ky * nx;
            c = ((iz0 + kz) * gg.g + (iy0 + ky)) * gg.g + (ix0 + kx);
            const uint4 cr = s_cell[c];
            const float dx = __builtin_fmaxf(
                __builtin_fmaxf(half_bits_to_float(cr.x & 0xFFFFu) - qx, qx - half_bits_to_float(cr.x >> 16)), 0.f);
            const float dy = __builtin_fmaxf(
                __builtin_fmaxf(half_bits_to_float(cr.y & 0xFFFFu) - qy, qy - half_bits_to_float(cr.y >> 16)), 0.f);
            const float dz = __builtin_fmaxf(
                __builtin_fmaxf(half_bits_to_float(cr.z & 0xFFFFu) - qz, qz - half_bits_to_float(cr.z >> 16)), 0.f);
            const float tq = tm - __uint_as_float(cr.w);
            cpass = tq >= 0.f && sqdist3(dx, dy, dz) <= tq * tq;  // empty cell: inf
          }
          const unsigned long long cmask = __ballot(cpass);
          if (cpass) wl[nlist + __builtin_popcountll(cmask & ((1ull << lane) - 1ull))] = (unsigned short)c;
          nlist += __builtin_popcountll(cmask);
          cb += kWave;
          enum_done = cb >= nsub;
        }
        // (3) visit the listed cells, 16 per step (a 16-lane row takes 4 cells): 4 independent
        // coordinate loads per lane in flight, prices from LDS
#ifdef MVP_EMD_PROFILE
        const long long tv0 = __builtin_readcyclecounter();
        pcells += nlist;
#endif
        for (int k0 = 0; k0 < nlist; k0 += 16) {
          int s[4], s1[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int k = k0 + r * 4 + row;
            s[r] = 0;
            s1[r] = 0;
            if (k < nlist) {
              const int cc2 = wl[k];
              s[r] = c_start[cc2] + l16;
              s1[r] = c_start[cc2 + 1];
            }
          }
          bool more = true;
          while (more) {
            float4 o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = s[r] < s1[r] ? sc.obj[s[r]] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int r = 0; r < 4; ++r) consider(s[r] < s1[r], s[r], o[r]);
            bool again = false;  // cells with more than 16 members (rare): next 16
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              s[r] += 16;
              again |= s[r] < s1[r];
            }
            more = __any(again);
          }
        }
        nlist = 0;
#ifdef MVP_EMD_PROFILE
        tvis += __builtin_readcyclecounter() - tv0;
#endif
        if (enum_done) break;
      }

      // (4) merge the lanes: exact best / second best with the reference's tie order
      top2k_merge_step<0xB1, 0xF>(lb1, lbk, lb2, lb2k, n, tpu, sc.perm);
      top2k_merge_step<0x4E, 0xF>(lb1, lbk, lb2, lb2k, n, tpu, sc.perm);
      top2k_merge_step<0x141, 0xF>(lb1, lbk, lb2, lb2k, n, tpu, sc.perm);
      top2k_merge_step<0x140, 0xF>(lb1, lbk, lb2, lb2k, n, tpu, sc.perm);
      top2k_merge_step<0x142, 0xA>(lb1, lbk, lb2, lb2k, n, tpu, sc.perm);
      top2k_merge_step<0x143, 0xC>(lb1, lbk, lb2, lb2k, n, tpu, sc.perm);
      const float b1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lb1), 63));
      const float b2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lb2), 63));
      int bk = __builtin_amdgcn_readlane(lbk, 63);
      int b2k = __builtin_amdgcn_readlane(lb2k, 63);
      if (bk < 0) {  // cannot happen (>= 2 objects always survive); never index with -1
        if (lane == 0) s_err = 1;
        bk = 0;
        b2k = -1;
      }

      // (5) rebuild the cache from the staged candidates
      if (caching) {
        float tau = b2 - delta;  // everything the search skipped is worth less (seed <= b2)
        int newcc = 0;
        if (nst <= kStage) {
          const bool have = lane < nst;
          const float v = have ? st_v[wave][lane] : -1e9f;
          const int slot = have ? st_slot[wave][lane] : 0;
          const float d = have ? st_d[wave][lane] : 0.f;
          const bool keep = have && v >= tau;
          const unsigned long long km = __ballot(keep);
          const int kc = __builtin_popcountll(km);
          int rank = __builtin_popcountll(km & ((1ull << lane) - 1ull));
          if (kc > kTailK) {  // the best kTailK stay (descending order; ties: lower lane first); the next bounds the rest
            rank = 0;
            unsigned long long mm = km;
            while (mm) {
              const int l = __builtin_ctzll(mm);
              mm &= mm - 1;
              const float vl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
              rank += (vl > v || (vl == v && l < lane)) ? 1 : 0;
            }
            const unsigned long long mk = __ballot(keep && rank == kTailK);
            tau = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), __builtin_ctzll(mk)));
          }
          if (keep && rank < kTailK) {
            e_cs[e][rank] = (unsigned short)slot;
            e_cd[e][rank] = d;
          }
          newcc = min(kc, kTailK);
        }
        if (lane == 0) {
          e_tau[e] = tau;
          e_i[e].w = newcc;
        }
      }
#ifdef MVP_EMD_PROFILE
      if (lane == 0) {
        atomicAdd(&s_prof[1], 1ull);
        atomicAdd(&s_prof[3], (unsigned long long)(__builtin_readcyclecounter() - tb0));
        atomicAdd(&s_prof[4], (unsigned long long)plin);
        atomicAdd(&s_prof[5], (unsigned long long)pnsub);
        atomicAdd(&s_prof[6], (unsigned long long)pcells);
        atomicAdd(&s_prof[8], (unsigned long long)phome);
        atomicAdd(&s_prof[11], (unsigned long long)tvis);
      }
#endif
      int drawn = 0;
      if (lane == 0) {
        const float inc = b1 - b2 + eps;
        if constexpr (W > 1) {
          publish(u, bk, inc, j, b2k);
        } else {
          s_bo[u] = bk;
          s_binc[u] = inc;
          s_bj[u] = j;
          s_b2k[u] = b2k;
        }
        drawn = atomicAdd(&s_next, 1);
      }
      mi = __builtin_amdgcn_readlane(drawn, 0);
    }
#ifdef MVP_EMD_PROFILE
    const long long tpg = __builtin_readcyclecounter();
    if (t == 0) s_prof[0] += (unsigned long long)(U - nmiss);
#endif

    // ---------------- all bids of the round.  W > 1: thread v polls record v of the cloud-wide
    // list (member by member, in the order everybody derives from the tracked counts) until both
    // words carry this round's tag.
    int off_me = 0;
    int r_o = 0, r_j = -1, r_prev = -1, r_w = 0;
    float r_inc = 0.f;
    const bool r_mine = t < Utot;
    if constexpr (W > 1) {
      int cnt[W];
#pragma unroll
      for (int w = 0; w < W; ++w) {
        cnt[w] = s_mcnt[w];
        if (w < wg) off_me += cnt[w];
      }
      if (r_mine) {
        int i = t;
#pragma unroll
        for (int ww = 0; ww < W - 1; ++ww)
          if (r_w == ww && i >= cnt[ww]) {
            i -= cnt[ww];
            r_w = ww + 1;
          }
        const u64 *p = xb_at(it & 1, r_w, i);
        u64 a = 0ull, bb = 0ull;
        bool ok = false;
        for (unsigned spins = 0; spins < kSpinLimit; ++spins) {
          a = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bb = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = (unsigned)(a >> 32) == tag && (unsigned)(bb >> 48) == (tag & 0xFFFFu);
          if (ok) break;
          __builtin_amdgcn_s_sleep(8);   // ~0.5k cycles: hundreds of pollers must not crowd the searches' loads out of L2
        }
        if (!ok) s_abort = 1;
        r_o = (int)((unsigned)a & 0xFFFFu);
        r_j = (int)(((unsigned)a >> 16) & 0xFFFFu);
        r_inc = __uint_as_float((unsigned)bb);
        s_bo[t] = r_o;
        s_bj[t] = r_j;
        s_binc[t] = r_inc;
        s_b2k[t] = (int)((unsigned)(bb >> 32) & 0xFFFFu) - 1;
        s_bm[t] = (unsigned char)r_w;
      } else if (t >= kEmdThreads - W) {  // the last W threads wait for the members' heartbeats
        const u64 *p = xb_at(it & 1, kEmdThreads - 1 - t, kTailCap - 1);
        bool ok = false;
        for (unsigned spins = 0; spins < kSpinLimit; ++spins) {
          ok = (unsigned)(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == tag;
          if (ok) break;
          __builtin_amdgcn_s_sleep(8);
        }
        if (!ok) s_abort = 1;
      }
    } else {
      lds_barrier();  // the phase B bids are in the LDS arrays
      if (r_mine) {
        r_o = s_bo[t];
        r_j = s_bj[t];
        r_inc = s_binc[t];
      }
    }
    // ---------------- GetMax (emd_cuda.cu:181-194), exactly, for ALL bids of the cloud
    bool r_win = true;
    if constexpr (LOWN) {
      if (r_mine) {
        r_prev = s_owner[r_o];           // owner before this round (0xFFFF: free)
        s_gcnt[t] = 0;
        s_gmax[t] = 0u;
      }
      lds_barrier();
      if (s_abort) {  // a member stopped publishing (tens of seconds): give up, loudly
        aborted = true;
        break;
      }
#ifdef MVP_EMD_PROFILE
      const long long tp1 = __builtin_readcyclecounter();
      cyc_fetch += tp1 - tpg;
#endif
      if (r_mine) s_owner[r_o] = (unsigned short)(0x8000u | (unsigned)t);  // person ids are < 0x8000
      lds_barrier();
      int rep = 0;
      if (r_mine) {
        rep = s_owner[r_o] & 0x7FFF;
        atomicAdd(&s_gcnt[rep], 1);
        atomicMax(&s_gmax[rep], emd_f2ord(r_inc));
      }
      lds_barrier();
      if (r_mine && s_gcnt[rep] > 1) {  // rare: several bids on this object
        const float mxi = emd_ord2f(s_gmax[rep]);
        r_win = emd_in_band(r_inc, mxi);
        if (r_win)
          for (int v = 0; v < Utot; ++v)
            if (s_bo[v] == r_o && s_bj[v] > r_j && emd_in_band(s_binc[v], mxi)) r_win = false;
      }
#ifdef MVP_EMD_PROFILE
      cyc_resolve += __builtin_readcyclecounter() - tp1;
#endif
    } else {
      if (t >= Utot && t < Utot + 4 && t < kTailCap) s_bo[t] = -2;  // padding of the scan
      lds_barrier();
      if (r_mine) {
        // branch-free scan, four bids per LDS read
        float mxi = -1e9f;
        int same = 0;
        const int4 *bo4 = reinterpret_cast<const int4 *>(s_bo);
        const float4 *bi4 = reinterpret_cast<const float4 *>(s_binc);
        const int n4 = (Utot + 3) >> 2;
#pragma unroll 4
        for (int v = 0; v < n4; ++v) {
          const int4 ov = bo4[v];
          const float4 iv = bi4[v];
          same += (ov.x == r_o) + (ov.y == r_o) + (ov.z == r_o) + (ov.w == r_o);
          mxi = __builtin_fmaxf(mxi, ov.x == r_o ? iv.x : -1e9f);
          mxi = __builtin_fmaxf(mxi, ov.y == r_o ? iv.y : -1e9f);
          mxi = __builtin_fmaxf(mxi, ov.z == r_o ? iv.z : -1e9f);
          mxi = __builtin_fmaxf(mxi, ov.w == r_o ? iv.w : -1e9f);
        }
        if (same > 1) {  // rare: several bids on this object
          r_win = emd_in_band(r_inc, mxi);
          if (r_win)
            for (int v = 0; v < Utot; ++v)
              if (s_bo[v] == r_o && s_bj[v] > r_j && emd_in_band(s_binc[v], mxi)) r_win = false;
        }
        r_prev = ((s_owned[r_o >> 5] >> (r_o & 31)) & 1u) ? 0 : 0xFFFF;  // W == 1: only "owned or free" is known here
      }
      lds_barrier();  // (every scan is done before owner bits / prices change)
    }
    if (last) r_win = true;
    const bool r_evict = r_mine && r_win && !last && r_prev != 0xFFFF;
#ifdef MVP_EMD_PROFILE
    const long long tpr = __builtin_readcyclecounter();
#endif

    // ---------------- Assign (emd_cuda.cu:196-215), by the bid's own thread.  A bid of this
    // member also moves its pool entry; the loads of an evicted person's record and cache are
    // issued first so that they travel while the prices are updated.
    const bool a_own = r_mine && r_w == wg;
    const int a_u = t - off_me;
    int a_e = 0, a_prev = -1;
    v4u ch[7] = {}, pa = {0u, 0u, 0u, 0u}, pb = {0u, 0u, 0u, 0u};
    if (a_own) {
      a_e = s_be[a_u];
      if (r_evict) {
        if constexpr (LOWN) a_prev = r_prev;
        else a_prev = ld_i32(&sc.ostate[r_o].z);
        const unsigned pvc = off_cache + (unsigned)a_prev * kCacheRec;
        if (caching) {
#pragma unroll
          for (int c = 0; c < 7; ++c) ch[c] = ldg16(pvc + (unsigned)c * 16u);  // 0-1 slots, 2-5 distances, 6 {tau, count}
        }
        pa = ldg16(off_person + (2u * (unsigned)a_prev) * 16u);
        pb = ldg16(off_person + (2u * (unsigned)a_prev + 1u) * 16u);
      }
    }
    // every price rise (and new owner) of the round goes into this member's own copy
    if (r_mine && r_win && !last) {
      if (r_prev == 0xFFFF) {  // a free object gets its first owner: one unassigned person fewer
        atomicAdd(&s_nfree, 1);
        atomicSub(&s_mcnt[r_w], 1);
        if constexpr (!LOWN) atomicOr(&s_owned[r_o >> 5], 1u << (r_o & 31));
      }
      if constexpr (LOWN) s_owner[r_o] = (unsigned short)r_j;
      const float pold = s_price[r_o];
      const float pnew = pold + r_inc;
      s_price[r_o] = pnew;
      // the object's cell: the last c with c_start[c] <= o; its lower bound only when (one of)
      // its cheapest members got dearer
      int lo = 0, hi = ncell;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int)c_start[mid] <= r_o) lo = mid;
        else hi = mid;
      }
      float *lbp = reinterpret_cast<float *>(&s_cell[lo]) + 3;
      if (pold <= *lbp) {
        float pm = pnew;
        const int e1 = c_start[lo + 1];
        for (int s2 = c_start[lo]; s2 < e1; ++s2) pm = __builtin_fminf(pm, s2 == r_o ? pm : s_price[s2]);
        *lbp = pm;
      }
    }
    if (a_own) {
      if (last) {
        st_i32(&ass[r_j], r_o);
      } else if (r_win) {
        // the winner leaves the pool: its cache and hints go to scratch
        const unsigned myc = off_cache + (unsigned)r_j * kCacheRec;
        if (caching) {
          const v4u *cs = reinterpret_cast<const v4u *>(&e_cs[a_e][0]);
          const v4u *cd = reinterpret_cast<const v4u *>(&e_cd[a_e][0]);
          stg16(cs[0], myc);
          stg16(cs[1], myc + 16u);
          stg16(cd[0], myc + 32u);
          stg16(cd[1], myc + 48u);
          stg16(cd[2], myc + 64u);
          stg16(cd[3], myc + 80u);
          v4u r;
          r.x = __float_as_uint(e_tau[a_e]);
          r.y = (unsigned)e_i[a_e].w;
          r.z = 0u;
          r.w = 0u;
          stg16(r, myc + 96u);
        }
        if (r_evict) st_i32(&ass[a_prev], -1);
        if constexpr (!LOWN) st_i32(&sc.ostate[r_o].z, r_j);
        st_i32(&ass[r_j], r_o);
        v4u hi;
        hi.x = (unsigned)r_o;
        hi.y = (unsigned)r_o;
        hi.z = (unsigned)s_b2k[t];
        hi.w = __float_as_uint(r_inc);
        stg16(hi, off_person + (2u * (unsigned)r_j + 1u) * 16u);
        // the entry: handed to the evicted person, or dead
        if (r_evict) {
          if (caching) {
            v4u *cs = reinterpret_cast<v4u *>(&e_cs[a_e][0]);
            v4u *cd = reinterpret_cast<v4u *>(&e_cd[a_e][0]);
            cs[0] = ch[0];
            cs[1] = ch[1];
            cd[0] = ch[2];
            cd[1] = ch[3];
            cd[2] = ch[4];
            cd[3] = ch[5];
            e_tau[a_e] = __uint_as_float(ch[6].x);
          }
          e_q[a_e] = make_float4(__uint_as_float(pa.x), __uint_as_float(pa.y), __uint_as_float(pa.z), 0.f);
          e_i[a_e] = make_int4(a_prev, (int)pb.y, (int)pb.z, caching ? (int)ch[6].y : 0);
        } else {
          e_live[a_e] = 0;
        }
      } else {
        // lost: keeps its entry; this bid's best / second best seed the next one
        e_i[a_e].y = r_o;
        e_i[a_e].z = s_b2k[t];
      }
    }
    // W == 1: later rounds re-read what this one stored (records, caches) through the same
    // CU -- wait for the stores; W > 1: the __syncthreads after the next phase A does
    if constexpr (W == 1) __syncthreads();
    else lds_barrier();
#ifdef MVP_EMD_PROFILE
    const long long tp2 = __builtin_readcyclecounter();
#endif

    // ---------------- next round's order list: the live entries, by wave 0
    if (wave == 0) {
      int base = 0;
#pragma unroll
      for (int h = 0; h < POOL / kWave; ++h) {
        const bool live = e_live[h * kWave + lane] != 0;
        const unsigned long long lm = __ballot(live);
        if (live) s_order[base + __builtin_popcountll(lm & ((1ull << lane) - 1ull))] = (unsigned short)(h * kWave + lane);
        base += __builtin_popcountll(lm);
      }
      if (lane == 0) {
        s_U = base;
        s_next = kEmdWaves;
        s_nmiss = 0;
        if (base != s_mcnt[wg]) s_err = 1;  // cannot happen: the tracked count is this member's pool
      }
    }
    lds_barrier();
    U = s_U;
    Utot = U0 - s_nfree;   // (cumulative)
#ifdef MVP_EMD_PROFILE
    const long long tp3 = __builtin_readcyclecounter();
    cyc_a += tpa - tp0;
    cyc_b += tpg - tpa;
    cyc_assign += tp2 - tpr;
    cyc_compact += tp3 - tp2;
#endif
  }
#ifdef MVP_EMD_PROFILE
  if (t == 0 && cloud < 2) {
    printf("tail cloud %d wg %d/%d: rounds %lld own bids %lld | cycles setup %lld phaseA %lld phaseB %lld fetch %lld resolve %lld assign %lld compact %lld\n",
           cloud, wg, W, n_rounds, n_bids, cyc_setup, cyc_a, cyc_b, cyc_fetch, cyc_resolve, cyc_assign, cyc_compact);
    printf("tail cloud %d wg %d: hits %llu misses %llu (%llu cycles each, visits %llu) linear %llu mean sub-box %llu cells visited %llu home-seeds %llu\n",
           cloud, wg, s_prof[0], s_prof[1], s_prof[3] / (s_prof[1] + 1), s_prof[11] / (s_prof[1] + 1), s_prof[4],
           s_prof[5] / (s_prof[1] + 1), s_prof[6] / (s_prof[1] + 1), s_prof[8]);
  }
#endif

  if (aborted) {
    // A cluster wait ran into its bound (members not co-resident for tens of seconds): fail
    // loudly -- NaN distances, -1 assignments, negative status.
    if (t == 0) stats[0] = -2;
    for (int j = t; j < n; j += kEmdThreads) {
      dist[j] = __builtin_nanf("");
      ass[j] = -1;
    }
    return;
  }
  if (t == 0) {
    if (wg == 0) stats[0] = s_err ? -1 : stats[0] + n_rounds;
    atomicAdd(reinterpret_cast<unsigned long long *>(&stats[1]), (unsigned long long)n_bids);
  }
  // ---------------- CalcDist (emd_cuda.cu:217-226); slots -> object indices.  The other
  // members' last assignments must have landed: one more all-gather.
  if constexpr (W > 1) {
    if (!emd_cluster_gather<W>(slots, wg, ++epoch, &s_U, &s_err, s_gout, &s_abort)) {
      if (t == 0) stats[0] = -2;
      for (int j = t; j < n; j += kEmdThreads) {
        dist[j] = __builtin_nanf("");
        ass[j] = -1;
      }
      return;
    }
    bool err = false;
#pragma unroll
    for (int w = 0; w < W; ++w) err |= s_gout[2 * w + 1] != 0u;
    if (err && wg == 0 && t == 0) stats[0] = -1;
  } else {
    __syncthreads();
  }
  const int share = n / W;  // n % 1024 == 0
  for (int j = wg * share + t; j < (wg + 1) * share; j += kEmdThreads) {
    const int s = ld_i32(&ass[j]);
    const float4 o = sc.obj[s];  // coordinates never change
    const float dx = xyz1[j * 3 + 0] - o.x;
    const float dy = xyz1[j * 3 + 1] - o.y;
    const float dz = xyz1[j * 3 + 2] - o.z;
    dist[j] = sqdist3(dx, dy, dz);
    ass[j] = sc.perm[s];
  }
}

template <int W>
static hipError_t emd_tail_launch_w(int b, int n, const float *xyz1, float *dist, int *assignment, float eps, int iters,
                                    char *scratch, float delta, hipStream_t stream) {
  int bpad = W == 1 ? b : (b + 7) / 8 * 8;
  if (W == 1) {
    hipLaunchKernelGGL(emd_tail_kernel<1>, dim3(b), dim3(kEmdThreads), 0, stream, b, bpad, n, xyz1, dist, assignment,
                       eps, iters, scratch, delta);
    return hipSuccess;
  }
  // members wait for each other: cooperative launch (residency is checked)
  void *args[] = {&b, &bpad, &n, &xyz1, &dist, &assignment, &eps, &iters, &scratch, &delta};
  return hipLaunchCooperativeKernel(reinterpret_cast<const void *>(emd_tail_kernel<W>), dim3(W * bpad),
                                    dim3(kEmdThreads), args, 0, stream);
}

// w = workgroups per cloud (1, 2, 4, 8); falls back to 1 if the cluster does not fit the device
void emd_tail_launch(int b, int n, int w, const float *xyz1, float *dist, int *assignment, float eps, int iters,
                     char *scratch, float delta, hipStream_t stream) {
  hipError_t err = hipErrorUnknown;
  if (w == 8) err = emd_tail_launch_w<8>(b, n, xyz1, dist, assignment, eps, iters, scratch, delta, stream);
  else if (w == 4) err = emd_tail_launch_w<4>(b, n, xyz1, dist, assignment, eps, iters, scratch, delta, stream);
  else if (w == 2) err = emd_tail_launch_w<2>(b, n, xyz1, dist, assignment, eps, iters, scratch, delta, stream);
  if (err != hipSuccess) {
    (void)hipGetLastError();
    (void)emd_tail_launch_w<1>(b, n, xyz1, dist, assignment, eps, iters, scratch, delta, stream);
  }
}

}  // namespace mvp
