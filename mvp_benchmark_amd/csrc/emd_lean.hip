// Lean tail kernel of the EMD auction for gfx950: the one-bidder-per-wave rounds.
//
// emd.hip's clustered kernel (emd_auction_kernel<W>) runs the auction's first rounds, where a
// workgroup has more bidders than waves and four bidders share a wave.  Once every workgroup of
// a cloud is below that threshold (round ~100 of 3000 at the headline shape; the number of
// unassigned persons never grows) it hands the cloud over -- lists of unassigned persons, round
// counter, grid geometry -- and the kernels of THIS file run the remaining rounds (95 % of them)
// with the same two all-gathers per round and the same exact, lossless search (see emd.hip for the
// design notes, the proofs and the reference citations: utils/metrics/EMD/emd_cuda.cu:95-215).
// What they do not carry: the four-bidders-per-wave schedule, the grid build, lists longer than the
// LDS record cache -- so the register allocator works for the one path that matters here.  Results
// are bit-identical to running every round in emd_auction_kernel (and to the oracle): the state a
// launch resumes from is the complete auction state (prices, owners, bid hints, assignment) in the
// per-cloud scratch.
//   emd_lean_body<W>        the rounds of one cloud on a cluster of W workgroups (device function)
//   emd_lean_kernel<W>      every cloud on the cluster width the first kernel ran with
//   emd_lean_tiers_kernel   from round 300 on: the workgroups dealt out again, 8 .. 2 per cloud by the
//                           clouds' load (a launch lasts as long as its slowest cloud; DESIGN.md 5.2, notebook 5e)
// The round loop of emd_lean_body is kept in four fragments included into it (round 5; one file had grown to 1766 lines):
//   emd_lean_bid.inc             Bid: one wave per bidder (both round shapes)
//   emd_lean_round_plain.inc     plain rounds: bid atomics, all-gather, GetMax, Assign
//   emd_lean_round_gathered.inc  gathered-bid rounds: bid records, one cluster-wide wait, every member settles every bid
//   emd_lean_round_end.inc       list sizes, hand-over / stop checks, price bounds, entry into the gathered rounds
#include <cstdlib>
#include <type_traits>

#include "emd_common.h"
#include "emd_index.h"
#include "emd_resident.h"

namespace mvp {

constexpr int kLeanBid = 512;

// emd_resident.hip
hipError_t emd_resident_launch(int b, int n, float *dist, int *assignment, float eps, int iters, char *scratch,
                               hipStream_t stream);

// The kernels' LDS, one object per kernel: the tiers kernel holds the round loop three times (one
// instance per cluster width), all on the same memory.
struct LeanShared {
  // (order: what the round loop addresses with an immediate offset comes first -- a DS instruction's
  // offset field covers 64 KB)
  float4 l_lo[kMaxLeaves], l_hi[kMaxLeaves];   // the leaves' boxes + price bounds (emd_index.h)
  float4 n_lo[kMaxNodes], n_hi[kMaxNodes];     // the nodes'
  int s_cnt[2];
  int s_next;
  int s_err, s_abort, s_nchg, s_xcc;
  int s_alarm[2];
  unsigned s_gout[2 * kMaxCluster];
  int s_bj[kLeanBid], s_bo[kLeanBid], s_b2k[kLeanBid];
  float s_binc[kLeanBid];
  int s_own_chg[kLeanBid];
  // (gathered-bid rounds: a list holds <= kGCap = kRecCap / 2 records, and the upper halves of these four 8 KB arrays
  // hold the round's bids per object -- one 8-bit counter each, 4 x 4 KB for <= 16384 objects; emd_bid_count below)
  float4 s_rq[2][kRecCap];
  int4 s_ri[2][kRecCap];
  unsigned short w_list[kEmdWaves][4 * kRowListCap];
  // gathered-bid rounds: every member's view of the cloud -- who owns which object, the round's bids,
  // every member's list length
  unsigned short s_owner[kGMaxN];     // object slot -> person (0xFFFF: free)
  unsigned short s_go[kGCap], s_gj[kGCap];   // the round's bids by position in the cloud-wide order: object, bidder
  float s_ginc[kGCap];                //   ... increment
  unsigned short s_won[2][kGCap];     // leaves of the objects won this round (their price bound is re-scanned)
  int s_nwon[2];
  int s_pub;                          // bids this member has published this round
  int s_gc[3][kMaxCluster];           // every member's list length: current round / next / (being zeroed)
  // rounds of at most 16 bidders on member 0 alone (emd_lean_round_few.inc): bids per bucket of object slots by round
  // parity, bidders of the round by round % 3
  alignas(16) int f_cnt[2][256];
  int f_act[3];
#ifdef MVP_EMD_PROFILE
  int s_wbusy[kEmdWaves];
  unsigned long long s_hist2[4];
  float s_loose[2][3];
  unsigned long long s_slow[2][8];
  unsigned long long s_hist[16];
#endif
};

// The remaining rounds of one cloud on a cluster of WB workgroups, of which this is member `wg`.
// it_stop < iters: stop before round it_stop and leave the lists for the next launch, as the first
// kernel left them for this one (`which`: the set of barrier granules this launch uses; launches
// that share a set go on counting its barriers).  The previous launch's cluster may have
// had another width: entry p of its lists' concatenation goes to member p % WB.  Which workgroups
// serve a cloud does not change a bit of the result -- the auction state is in the scratch, the
// bids of a round do not depend on each other or on their order.
// Returns 1 when this workgroup is to finish the cloud LDS-resident itself (`fused`: no launch in between; the
// members of the cluster have left their lists in the scratch and all but member 0 are gone), else 0.
template <int WB>
__device__ __forceinline__ int emd_lean_body(LeanShared &sh, const int cloud, const int wg, int b,
                                             int n, const float *__restrict__ xyz1, float *__restrict__ dist,
                                             int *assignment, float eps, int iters, char *scratch, int fast_ok,
                                             int it_stop, int which, int u_stop, int fused = 0) {
  constexpr int W = WB, WM = WB;
  const int t = threadIdx.x;
  const int lane = t & (kWave - 1);
  const int wave = t >> 6;   // (marking it wave-uniform with readfirstlane: 3 VGPRs fewer, the same 34.5-34.8 ms at the headline)
  char *tail = scratch + (size_t)b * emd_scratch_per_cloud(n);
  if (cloud >= b) return 0;
  char *cbase = scratch + (size_t)cloud * emd_scratch_per_cloud(n);
  u64 *slots = emd_granules(tail, b, cloud, which);   // this launch's own granules (zeroed by the host)
  EmdHandover *resume = emd_handover(tail, b, cloud);
  long long *stats = emd_stats(tail, b, cloud);
  const int it0 = resume->next_it;
  if (it0 == 0 || it0 >= it_stop) return 0;   // finished already / not this launch's rounds (uniform over the cluster)
  // u_stop > 0: the cloud is left to the resident kernel (emd_resident.hip) as soon as at most u_stop persons
  // are unassigned -- possibly at once: the record and the lists stay as the previous launch wrote them
  if (resume->utot <= u_stop && iters - it0 >= kResMinRounds) return fused && wg == 0 ? 1 : 0;
  xyz1 += (size_t)cloud * n * 3;
  dist += (size_t)cloud * n;
  int *ass = assignment + (size_t)cloud * n;
  const EmdScratch sc = emd_carve(cbase, n);

  // ---- accessors of the shared auction state.  W == 1: plain.  W > 1: loads
  // bypass this CU's L1 (sc1), stores are written through (sc1), so data is
  // visible to the other workgroups once the store is acknowledged.
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(cbase, 0, (int)emd_scratch_per_cloud(n), 0x00020000);
  auto ld_obj = [&](int s) -> float4 {
    if constexpr (WB == 1) {
      return sc.obj[s];
    } else {
      const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)s * 16u, 0, 16);
      return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
  };
  auto ld_price = [&](int s) -> float {  // obj[s].w alone
    if constexpr (WB == 1) return sc.obj[s].w;
    else return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)s * 16u + 12u, 0, 16));
  };
  auto ld_ostate = [&](int s) -> int4 {
    if constexpr (WB == 1) {
      return sc.ostate[s];
    } else {
      const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, ((unsigned)n + (unsigned)s) * 16u, 0, 16);
      return make_int4((int)v.x, (int)v.y, (int)v.z, (int)v.w);
    }
  };
  auto ld_person = [&](int j, int half) -> float4 {  // half 0 = lo, 1 = hi
    if constexpr (WB == 1) {
      return sc.person[2 * j + half];
    } else {
      const v4u v =
          __builtin_amdgcn_raw_buffer_load_b128(rs, (2u * (unsigned)n + 2u * (unsigned)j + (unsigned)half) * 16u, 0, 16);
      return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
  };
  // Stores of shared words.  When every member of the cluster was observed on
  // the same XCD (same_xcd, below) they share one L2, the coherence point of
  // that XCD's CUs: a plain store (L1 is write-through) is visible to the
  // others' L1-bypassing loads as soon as it is acknowledged, and the line
  // stays in L2 instead of being written through to memory and dropped.
  bool same_xcd = false;
  auto st_person_hi = [&](int j, int bid, int p1, int p2, float inc) {
    if constexpr (WB == 1) {
      sc.person[2 * j + 1] = make_float4(__int_as_float(bid), __int_as_float(p1), __int_as_float(p2), inc);
    } else {
      v4u v;
      v.x = (unsigned)bid; v.y = (unsigned)p1; v.z = (unsigned)p2; v.w = __float_as_uint(inc);
      if (same_xcd) __builtin_amdgcn_raw_buffer_store_b128(v, rs, (2u * (unsigned)n + 2u * (unsigned)j + 1u) * 16u, 0, 0);
      else __builtin_amdgcn_raw_buffer_store_b128(v, rs, (2u * (unsigned)n + 2u * (unsigned)j + 1u) * 16u, 0, 16);
    }
  };
  auto st_ostate = [&](int s, int owner) {  // key = 0 (no bid), new owner
    if constexpr (WB == 1) {
      sc.ostate[s] = make_int4(0, 0, owner, 0);
    } else {
      v4u v;
      v.x = 0u; v.y = 0u; v.z = (unsigned)owner; v.w = 0u;
      if (same_xcd) __builtin_amdgcn_raw_buffer_store_b128(v, rs, ((unsigned)n + (unsigned)s) * 16u, 0, 0);
      else __builtin_amdgcn_raw_buffer_store_b128(v, rs, ((unsigned)n + (unsigned)s) * 16u, 0, 16);
    }
  };
  auto st_i32 = [&](int *p, int v) {
    if constexpr (WB == 1) *p = v;
    else if (same_xcd) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto st_f32 = [&](float *p, float v) {
    if constexpr (WB == 1) *p = v;
    else if (same_xcd) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto ld_key = [&](int s) -> u64 {
    u64 *p = reinterpret_cast<u64 *>(&sc.ostate[s]);
    if constexpr (WB == 1) return *p;
    else return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };


  auto &l_lo = sh.l_lo; auto &l_hi = sh.l_hi; auto &n_lo = sh.n_lo; auto &n_hi = sh.n_hi;
  auto &w_list = sh.w_list;  // surviving leaves of a wave's search
  auto &s_cnt = sh.s_cnt;
  auto &s_next = sh.s_next;             // next undrawn list position of the round
  auto &s_err = sh.s_err; auto &s_abort = sh.s_abort; auto &s_nchg = sh.s_nchg; auto &s_xcc = sh.s_xcc;
  auto &s_alarm = sh.s_alarm;  // by round parity: set in Bid, read after the barrier, cleared a round later
  auto &s_gout = sh.s_gout;
#ifdef MVP_EMD_PROFILE
  auto &s_wbusy = sh.s_wbusy;
  auto &s_hist2 = sh.s_hist2;
  if (threadIdx.x < 4) s_hist2[threadIdx.x] = 0;
  auto &s_loose = sh.s_loose;  // [slow][sum of (seed threshold - final threshold) in cell widths, sum of seed threshold, evicted (no bid last round) count]
  if (threadIdx.x < 6) s_loose[threadIdx.x / 3][threadIdx.x % 3] = 0.f;
  auto &s_slow = sh.s_slow;  // [d >= 10k cycles][count, nsub, cells, visit steps, extra member iterations, folds, seed cycles, visit cycles]
  if (threadIdx.x < 16) s_slow[threadIdx.x >> 3][threadIdx.x & 7] = 0;
  auto &s_hist = sh.s_hist;  // bids: [0..7] duration buckets, [8] sum nsub, [9] sum cells visited, [10] count, [11] linear scans, [12] sum cycles
  if (threadIdx.x < 16) s_hist[threadIdx.x] = 0;
#endif
  // this round's bids, by list position
  auto &s_bj = sh.s_bj; auto &s_bo = sh.s_bo; auto &s_b2k = sh.s_b2k;
  auto &s_binc = sh.s_binc;
  auto &s_own_chg = sh.s_own_chg;   // leaves whose cheapest member this workgroup's winners made dearer this round
  // person records {qx,qy,qz,-} / {j, prev1, prev2, -} of the current / next unassigned list
  auto &s_rq = sh.s_rq;
  auto &s_ri = sh.s_ri;
  static_assert(kLeanBid == kRecCap, "one LDS slot per list position");
  // Bids per object of the round (gathered-bid rounds): byte counter of object o.  Exact -- a count of one means nobody
  // else bid for the object; 256 bids on one object wrap to 0 and a carry into the neighbour: both read "not one" and
  // take the (rare) path that looks at the bids themselves.  Kept at zero between rounds by their readers.
  static_assert(2 * kGCap <= kRecCap && kGMaxN <= 4 * 4096 && sizeof(float4) * kRecCap == 8192, "the counters' home");
  auto bid_count = [&](int o) -> unsigned char * {
    return reinterpret_cast<unsigned char *>(&s_rq[0][0]) + 4096 + ((o >> 12) << 13) + (o & 4095);
  };
  static_assert(sizeof(LeanShared) <= 160 * 1024, "one workgroup per CU: all of its LDS");
  auto &s_owner = sh.s_owner; auto &s_go = sh.s_go; auto &s_gj = sh.s_gj; auto &s_ginc = sh.s_ginc;
  auto &s_won = sh.s_won; auto &s_nwon = sh.s_nwon; auto &s_gc = sh.s_gc; auto &s_pub = sh.s_pub;
  u64 *const bid_area = emd_bid_area(tail, b, cloud);

  // ------------------------------------------------------------ resume
  const int lshift = emd_leaf_shift(n), nleaf = n >> lshift, nnode = (nleaf + kNodeFan - 1) / kNodeFan;
  const int kch = 1 << (lshift - 4);   // 16-slot chunks per leaf
  if (t == 0) {
    s_err = resume->err;
    s_abort = 0;
    s_alarm[0] = 0;
    s_alarm[1] = 0;
    s_next = kEmdWaves;
    s_nchg = 0;
  }
  // exact bounding box and exact price lower bound per leaf and node (ends with a barrier)
  emd_index_boxes(l_lo, l_hi, n_lo, n_hi, sc.obj, n, lshift);
  // this member's unassigned list, as the first kernel left it
  int *my_ulist = sc.ulist + (size_t)wg * 2 * n;
  {
    // The previous launch left `nlists` lists (its cluster width); entry p of their concatenation
    // goes to member p % W of this cluster.
    const int nl = resume->nlists;
    int total = 0;
#pragma unroll
    for (int w = 0; w < kMaxCluster; ++w) total += w < nl ? resume->cnt[w] : 0;
    const int cnt0 = total > wg ? (total - wg + W - 1) / W : 0;
    if (t < cnt0) {
      int p = wg + t * W, k = -1;
#pragma unroll
      for (int w = 0; w < kMaxCluster; ++w) {
        const int cw = w < nl ? resume->cnt[w] : 0;
        if (k < 0 && p >= 0 && p < cw) k = sc.ulist[(size_t)w * 2 * n + p];
        p -= cw;
      }
      const float4 pa = sc.person[2 * k], pb = sc.person[2 * k + 1];
      s_rq[0][t] = pa;
      s_ri[0][t] = make_int4(k, __float_as_int(pb.y), __float_as_int(pb.z), 0);
    }
    if (t == 0) {
      s_cnt[0] = cnt0;
      s_cnt[1] = 0;
    }
  }
  unsigned epoch = (unsigned)resume->epoch;   // (launches that share a set of granules go on counting; 0 after the first kernel)
  if constexpr (WB != 1) {
    __syncthreads();
    if (t == 0) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      s_xcc = (int)(xcc & 0xFu) + 1;
    }
    const bool ok = emd_cluster_gather_n<WM>(slots, W, wg, ++epoch, &s_xcc, &s_xcc, s_gout, &s_abort);
    same_xcd = (fast_ok & 1) != 0;
#pragma unroll
    for (int w = 1; w < WM; ++w) same_xcd &= w >= W || s_gout[2 * w] == s_gout[0];
    if (!ok) {
      if (wg == 0 && t == 0) stats[0] = -2;
      for (int j = t; j < n; j += kEmdThreads) {
        dist[j] = __builtin_nanf("");
        ass[j] = -1;
      }
      return 0;
    }
  } else {
    __syncthreads();
  }

  // ------------------------------------------------------------ the auction
  int cur = 0;
  int Utot = resume->utot;  // unassigned persons of the whole cloud
  long long n_rounds = 0, n_bids = 0;
  bool aborted = false;
  // Once at most kSoloMax persons are left (their number never grows) one
  // workgroup bids for all of them in a single pass and the cluster's barriers
  // would only add latency: member 0 adopts the others' lists and carries on
  // alone (same code, workgroup barriers), the others leave.
  bool clustered = WB != 1;
  // Gathered-bid rounds (entered below once at most kGCap persons are unassigned; their number never grows): `eg`
  // counts them (tags of the bid granules), `goff` = this member's first position in the cloud-wide order of bids.
  bool gm_now = false;
  const bool gm_ok = WB != 1 && n <= kGMaxN && (fast_ok & 2) != 0;   // (fast_ok: bit 0 = same-XCD stores, bit 1 = gathered-bid rounds)
  unsigned eg = (unsigned)resume->epoch_g;
  int goff = 0, gi = 0;
#ifdef MVP_EMD_GMTIME
  // phase clock of the gathered-bid rounds as wave 0 of member 0 sees them (cycles, summed over the rounds; tools/emd_gm_times.py)
  long long gmt[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long gm_prev = 0;
#define GMT(i) { const long long now_ = __builtin_readcyclecounter(); gmt[i] += now_ - gm_prev; gm_prev = now_; }
#else
#define GMT(i)
#endif
  // The rounds of at most 16 bidders (emd_lean_round_few.inc) take a collapsed cluster's cloud when the leaves hold 16 slots
  // (n <= 16384 = the owner map's size) and this launch runs to the auction's end; a resident hand-over below 16 persons
  // -- a test knob -- keeps the plain rounds (they watch for it).  (A lambda, evaluated where it is needed: two more
  // scalars alive through the round loops cost the gathered-bid rounds 900 more SGPR spill moves.)
  auto few_ok = [&]() { return MVP_EMD_FEW != 0 && n <= kGMaxN && lshift == 4 && it_stop >= iters && (u_stop <= 0 || u_stop >= kFewMax); };
  constexpr int kLeanSoloMax = kFewMax > kSoloMax ? kFewMax : kSoloMax;   // persons at which the cluster collapses to member 0
  int stop_cnt = -1;   // >= 0: the loop ended before round it + 1 with this many entries in this member's next list
  bool stop_for_res = false;   // ... because at most u_stop persons are left (not because round it_stop is next)
#ifdef MVP_EMD_PROFILE
  long long prof_gap = 0, prof_prev4 = 0;
  long long prof_pg1 = 0, prof_drain = 0, prof_gather = 0, cyc_bid = 0, cyc_sync1 = 0, cyc_assign = 0, cyc_sync2 = 0, n_alarm = 0, n_rebal = 0, prof_u = 0, prof_a1 = 0, prof_an = 0, prof_a2 = 0, prof_a3 = 0, prof_a4 = 0;
#endif
#ifdef MVP_EMD_PROFILE
  const long long t_loop0 = __builtin_readcyclecounter();
#endif
#ifdef MVP_EMD_CLOUDTIME
  const long long ct0 = wall_clock64();
  int ct_u[6] = {0, 0, 0, 0, 0, 0};
  long long ct_t[6] = {0, 0, 0, 0, 0, 0}, ct_b[6] = {0, 0, 0, 0, 0, 0};
#endif
  // The round loop, instantiated twice -- GM = false: bid atomics + two all-gathers per round; GM = true: gathered-bid
  // rounds -- so that each mode gets its own register allocation (one loop with a run-time mode: 98 -> 124 VGPRs and
  // twice the scalar spills in BOTH modes).  Returns 0 when the rounds are over (or the workgroup has nothing left to
  // do: ret_early), 2 / 3 when the next round is to run in the other mode.
  int it = it0;
  bool ret_early = false;
  auto rounds = [&](auto gm_tag) -> int {
  constexpr bool GM = decltype(gm_tag)::value;
  constexpr bool gm = GM;
  for (; it < iters; ++it) {
    int sw = 0;   // 2: gathered-bid rounds from the next round on; 3: back to the plain rounds (member 0 alone)
    if (Utot == 0) break;
#ifdef MVP_EMD_PROFILE
    if (cloud == 0 && wg == 0 && t == 0 && (it == 25 || it == 50 || it == 100 || it == 150 || it == 250 || it == 500 || it == 750 ||
                                           it == 1000 || it == 1500 || it == 2000 || it == 2500 || it == iters - 1))
      printf("head cloud 0: round %d starts at %lld cycles, unassigned %d\n", it, __builtin_readcyclecounter() - t_loop0, Utot);
#endif
#ifdef MVP_EMD_CLOUDTIME
    {
      const int marks[6] = {150, 200, 300, 500, 1000, 2000};
#pragma unroll
      for (int q = 0; q < 6; ++q)
        if (it == marks[q]) { ct_u[q] = Utot; ct_t[q] = wall_clock64() - ct0; ct_b[q] = n_bids; }
    }
#endif
    // gathered-bid round: every member's list length (scalar registers), this member's first position in the
    // cloud-wide order, and the heartbeat -- "every store this member issued in earlier rounds is performed" (the
    // previous settle ended with a drain); the others' settle of THIS round waits for it
    const int gcur = gi, gnxt = gi == 2 ? 0 : gi + 1, gzero = gi == 0 ? 2 : gi - 1;   // (rotating: this round's lengths, the next round's, the set cleared meanwhile)
    if constexpr (WB != 1) {
      if constexpr (GM) {
        ++eg;
#ifdef MVP_EMD_GMTIME
        if (gm_prev) GMT(0) else gm_prev = __builtin_readcyclecounter();   // [0] end of the last round's bookkeeping -> this round's start
        gmt[15] += 1;
        GMT(14) GMT(14)   // [14] = two stamps back to back (calibration)
#endif
        goff = 0;
        {
          const int cv = s_gc[gcur][lane & (kMaxCluster - 1)];   // (one LDS read; the sum is scalar work)
#pragma unroll
          for (int w = 0; w + 1 < WM; ++w)
            if (w < wg) goff += __builtin_amdgcn_readlane(cv, w);
        }
#ifdef MVP_EMD_HBTOP
        if (t == 0) {
#else
        if (t == 0 && s_gc[gcur][wg] == 0) {   // (a member with bidders: its last bid of the round raises the word)
#endif
          u64 *hb = bid_area + (size_t)(eg & 1u) * kGStride + 2 * kGCap + wg;
          if (same_xcd) __hip_atomic_store(hb, (u64)eg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          else __hip_atomic_store(hb, (u64)eg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    const int U = gm ? s_gc[gcur][wg] : s_cnt[cur];  // this workgroup's bidders
    n_rounds += 1;
    n_bids += U;
    const bool last = it == iters - 1;
    // thread_per_unass of the reference (emd_cuda.cu:107-109): fixes the tie
    // order only.
    const int tpu = -Utot;   // (resolved inside emd_precedes: only equal values ever need it)

#ifdef MVP_EMD_PROFILE
    const long long tp0 = __builtin_readcyclecounter();
    if (prof_prev4) prof_gap += tp0 - prof_prev4;
#endif
    // A bid = one returning 64-bit atomic max on the object's key; the value
    // it returns is examined one bid later (or after the loop), so the wave
    // never waits for it.
    u64 pend_old = 0ull;
    float pend_inc = 0.f;
    bool alarm = false;
    {
#include "emd_lean_bid.inc"
    }
    const int nxt = cur ^ 1;
#ifdef MVP_EMD_PROFILE
    const long long tp1 = __builtin_readcyclecounter();
    long long tp2 = tp1, tp3 = tp1;
    if (lane == 0) s_wbusy[wave] = (int)(tp1 - tp0);
#endif
    if constexpr (!GM) {
#include "emd_lean_round_plain.inc"
    } else {
#include "emd_lean_round_gathered.inc"
    }
#include "emd_lean_round_end.inc"
    if constexpr (GM) { GMT(10) }   // [10] counts, stop checks, price bounds
    cur ^= 1;
    if constexpr (GM) gi = gnxt;
    // member 0 alone with at most a bidder per wave, no launch boundary ahead: the rounds of emd_lean_round_few.inc
    if (__builtin_expect(!clustered && Utot > 0 && Utot <= kFewMax && it + 1 < iters, 0) && few_ok()) sw = 4;
    if (sw) {
      ++it;
      return sw;
    }
  }
  return 0;
  };
  auto few = [&](const bool owners_in_lds) {
#include "emd_lean_round_few.inc"
  };
  for (;;) {
    int code;
    if constexpr (WB != 1) code = gm_now ? rounds(std::true_type{}) : rounds(std::false_type{});
    else code = rounds(std::false_type{});
    if (code == 2) gm_now = true;
    else if (code == 3) gm_now = false;
    else {
      if (code == 4) few(gm_now);   // (gm_now: the loop that ran last kept the owner map in LDS)
      break;
    }
  }
  if (ret_early) return 0;
#ifdef MVP_EMD_PROFILE
  if (clustered && !aborted) {  // cost of the bare all-gather
    const long long tg0 = __builtin_readcyclecounter();
    for (int g = 0; g < 256; ++g)
      if (!emd_cluster_gather_n<WM>(slots, W, wg, ++epoch, &s_nchg, &s_nchg, s_gout, &s_abort, same_xcd)) break;
    if (t == 0 && cloud < 2 && wg == 0)
      printf("cloud %d: bare cluster all-gather %lld cycles each (W = %d, same_xcd %d)\n", cloud,
             (__builtin_readcyclecounter() - tg0) / 256, W, (int)same_xcd);
  }
#endif
  if (stop_cnt >= 0) {
    // ---- stopped before round it_stop: leave the lists for the next launch, as the first kernel
    // left them for this one (the prices, owners and bid hints are in the scratch already).  The
    // record's address is derived again here (from a value the compiler cannot trace back) so that
    // nothing of it stays in registers through the round loop.
    int cloud2 = cloud;
    asm volatile("" : "+s"(cloud2));
    char *tail2 = scratch + (size_t)b * emd_scratch_per_cloud(n);
    EmdHandover *rs = emd_handover(tail2, b, cloud2);
    long long *st2 = emd_stats(tail2, b, cloud2);
    int *ul2 = emd_carve(scratch + (size_t)cloud2 * emd_scratch_per_cloud(n), n).ulist + (size_t)wg * 2 * n;
    const int nx = cur ^ 1;
    if (t < stop_cnt) ul2[t] = s_ri[nx][t].x;
    const int stop_it = it + 1;   // (the loop was left by `break`: `it` is the last round that ran)
    if (t == 0) {
      atomicAdd(reinterpret_cast<unsigned long long *>(&st2[1]), (unsigned long long)n_bids);
      if (s_err) rs->err = 1;
      rs->cnt[wg] = stop_cnt;
#ifdef MVP_EMD_CLOUDTIME
      if (wg == 0) sc.chg[(size_t)kMaxCluster * kChgCap - 64 + (min(stop_it, 3900) >> 6)] = ((u64)W << 48) | ((u64)Utot << 32) | (u64)(unsigned)(wall_clock64() - ct0);
#endif
      if (wg == 0) {
        rs->utot = Utot;
        rs->nlists = clustered ? W : 1;
        rs->epoch = (int)epoch;
        rs->epoch_g = (int)eg;
        rs->next_it = stop_it;
        atomicAdd(reinterpret_cast<unsigned long long *>(&st2[0]), (unsigned long long)n_rounds);
      }
    }
    if (fused && stop_for_res) {
      // The cloud goes on LDS-resident on member 0 at once: the members' lists and counts above are plain stores --
      // released (written back), one more cluster barrier, then member 0 reads them (and the auction state the
      // rounds wrote through) after dropping its L1; the other members leave.
      if (clustered) {
        __syncthreads();
        if (t == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        u64 *slots2 = emd_granules(tail2, b, cloud2, which);
        if (!emd_cluster_gather_n<WM>(slots2, W, wg, ++epoch, &s_nchg, &s_nchg, s_gout, &s_abort, same_xcd)) return 0;
      } else {
        __syncthreads();
      }
      if (wg != 0) return 0;
      if (t == 0) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
      __syncthreads();
      return 1;
    }
    return 0;
  }
#ifdef MVP_EMD_CLOUDTIME
  if (wg == 0 && t == 0)   // 100 MHz constant clock
    sc.chg[(size_t)kMaxCluster * kChgCap - 64 + 63] = ((u64)W << 48) | (u64)(unsigned)(wall_clock64() - ct0);
#endif
  if (aborted) {
    // A cluster wait ran into its bound (the members were not co-resident for
    // tens of seconds).  Fail loudly: NaN distances, -1 assignments.
    if (t == 0) stats[0] = -2;
#ifndef MVP_EMD_STUCKDUMP
    for (int j = t; j < n; j += kEmdThreads) {
      dist[j] = __builtin_nanf("");
      ass[j] = -1;
    }
#endif
    return 0;
  }
#ifdef MVP_EMD_GMTIME
  if (t == 0 && wg == 0) {
    for (int k = 0; k < 16; ++k) sc.chg[(size_t)kMaxCluster * kChgCap - 128 + k] = (u64)gmt[k];
    sc.chg[(size_t)kMaxCluster * kChgCap - 128 + 16] = (u64)W;
  }
#endif
  if (t == 0) {
    // rounds: added to the first kernel's count; an internal error drives the sum far below zero
    if (wg == 0) {
      atomicAdd(reinterpret_cast<unsigned long long *>(&stats[0]), (unsigned long long)n_rounds);
      resume->next_it = 0;   // finished: nothing for a later launch
      resume->epoch_g = (int)eg;   // (statistics: gathered-bid rounds of the call)
      resume->last_width = W + 16 * which;   // (which: 1 = the launch after the first kernel, 2 = the tiered launch)
    }
    if (s_err) atomicAdd(reinterpret_cast<unsigned long long *>(&stats[0]), (unsigned long long)(-(1ll << 40)));
    atomicAdd(reinterpret_cast<unsigned long long *>(&stats[1]), (unsigned long long)n_bids);
#ifdef MVP_EMD_PROFILE
    if (cloud < 2)
      printf("cloud %d wg %d per wave-mode bid: seed %llu cycles, visits %llu cycles in %.2f steps folding %.1f candidates, rest (enumeration, finish) %llu\n", cloud, wg,
             s_hist2[0] / (s_hist[10] + 1), s_hist2[1] / (s_hist[10] + 1), (double)s_hist2[2] / (double)(s_hist[10] + 1), (double)s_hist2[3] / (double)(s_hist[10] + 1),
             (s_hist[12] - s_hist2[0] - s_hist2[1]) / (s_hist[10] + 1));
    if (cloud < 2)
      printf("cloud %d wg %d price-bound refreshes after round 100: %llu\n", cloud, wg, s_hist2[0] >> 40);
    if (cloud == 0 && wg == 0)
      for (int k = 0; k < 2; ++k) {
        const double c = (double)s_slow[k][0] + 1e-9;
        printf("cloud 0 wg 0 searches %s 10k cycles: seed threshold %.3f cell widths, of which %.3f loose (seed - final)\n", k ? ">=" : "<",
               s_loose[k][1] / c, s_loose[k][0] / c);
        printf("cloud 0 wg 0 searches %s 10k cycles: %llu | mean sub-box %.0f cells, visited %.1f, visit steps %.2f, extra member iterations %.2f, folds %.1f, seed %.0f cycles, visits %.0f cycles\n",
               k ? ">=" : "<", s_slow[k][0], s_slow[k][1] / c, s_slow[k][2] / c, s_slow[k][3] / c, s_slow[k][4] / c, s_slow[k][5] / c, s_slow[k][6] / c, s_slow[k][7] / c);
      }
    if (cloud < 2)
      printf("cloud %d wg %d gathered rounds (thread 0, %lld): contest check %lld, settle body %lld, drain %lld, to barrier end %lld cycles per round; flagged %.2f contested %.3f per round\n", cloud, wg, prof_an,
             prof_a1 / (prof_an + 1), prof_a2 / (prof_an + 1), prof_a3 / (prof_an + 1), prof_a4 / (prof_an + 1), (double)(s_hist2[3] >> 32) / (double)(prof_an + 1), (double)(s_hist2[2] >> 32) / (double)(prof_an + 1));
    if (cloud < 2 && false)
      printf("cloud %d wg %d Assign (thread 0, %lld samples): loads done at %lld cycles, eviction handled at %lld (sum over winning rounds / all), body done at %lld, phase %lld\n", cloud, wg, prof_an, prof_a1 / (prof_an + 1), prof_a2 / (prof_an + 1), prof_a3 / (prof_an + 1), prof_a4 / (prof_an + 1));
    if (cloud < 2)
      printf("cloud %d wg %d tail rounds %llu: bidders/round %.1f, busiest wave %llu cycles/round, mean wave %llu\n", cloud, wg, s_hist[15],
             (double)prof_u / (double)(s_hist[15] + 1), s_hist[13] / (s_hist[15] + 1), s_hist[14] / (s_hist[15] + 1));
    if (cloud < 2)
      printf("cloud %d wg %d wave-mode bids after round 100: %llu, mean cycles %llu, mean sub-box cells %llu, mean cells visited %llu, linear %llu | <2k %llu <4k %llu <8k %llu <16k %llu <32k %llu <64k %llu <128k %llu more %llu\n",
             cloud, wg, s_hist[10], s_hist[12] / (s_hist[10] + 1), s_hist[8] / (s_hist[10] + 1), s_hist[9] / (s_hist[10] + 1), s_hist[11],
             s_hist[0], s_hist[1], s_hist[2], s_hist[3], s_hist[4], s_hist[5], s_hist[6], s_hist[7]);
    if (cloud < 2)
      printf("cloud %d wg %d: rounds %lld bids %lld alarms %lld rebalances %lld | cycles bid %lld sync1 %lld assign %lld sync2 %lld gap %lld\n",
             cloud, wg, n_rounds, n_bids, n_alarm, n_rebal, cyc_bid, cyc_sync1, cyc_assign, cyc_sync2, prof_gap);
    if (cloud < 2)
      printf("cloud %d wg %d: sync2 = store drain %lld + closing gather %lld + list bookkeeping %lld + bound fetch (rest)\n", cloud, wg, prof_drain, prof_gather, prof_pg1);
#endif
  }
  // ---------------- CalcDist (emd_cuda.cu:217-226); slots -> object indices
  __syncthreads();
  // CalcDist: member w takes the blocks w, w + W, ... of 1024 persons (a collapsed cluster: member 0 all)
  for (int j = (clustered ? wg : 0) * kEmdThreads + t; j < n; j += (clustered ? W : 1) * kEmdThreads) {
    int s;
    if constexpr (WB == 1) s = ass[j];
    else s = __hip_atomic_load(&ass[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float4 o = sc.obj[s];  // coordinates never change
    const float dx = xyz1[j * 3 + 0] - o.x;
    const float dy = xyz1[j * 3 + 1] - o.y;
    const float dz = xyz1[j * 3 + 2] - o.z;
    dist[j] = sqdist3(dx, dy, dz);
    ass[j] = sc.perm[s];
  }
  return 0;
}

// Every cloud on a cluster of WT workgroups (grid WT * bpad, laid out as the first kernel's).
// RESN > 0 (2048 / 4096, with u_stop > 0): a cloud of at most RESN points does not wait for a next launch once at
// most u_stop persons are unassigned -- member 0 of its cluster runs the resident rounds (emd_resident.h) at once,
// on this workgroup, with the LDS re-used.
template <int WT, int RESN>
__global__ __launch_bounds__(kEmdThreads) void emd_lean_kernel(
    int b, int bpad, int n, const float *__restrict__ xyz1, float *__restrict__ dist, int *assignment,
    float eps, int iters, char *scratch, int fast_ok, int it_stop, int which, int u_stop) {
  const int cloud = WT == 1 ? (int)blockIdx.x : (int)blockIdx.x % bpad;
  const int wg = WT == 1 ? 0 : (int)blockIdx.x / bpad;
  if constexpr (RESN == 0) {
    __shared__ LeanShared sh;
    emd_lean_body<WT>(sh, cloud, wg, b, n, xyz1, dist, assignment, eps, iters, scratch, fast_ok, it_stop, which, u_stop);
  } else {
    constexpr size_t kBytes = sizeof(LeanShared) > sizeof(ResShared<RESN>) ? sizeof(LeanShared) : sizeof(ResShared<RESN>);
    __shared__ __attribute__((aligned(16))) char raw[kBytes];
    if (emd_lean_body<WT>(*reinterpret_cast<LeanShared *>(raw), cloud, wg, b, n, xyz1, dist, assignment, eps, iters, scratch,
                          fast_ok, it_stop, which, u_stop, 1))
      emd_resident_body<RESN>(*reinterpret_cast<ResShared<RESN> *>(raw), cloud, b, n, dist, assignment, eps, iters, scratch);
  }
}

// TIERED widths (grid 4 * bpad, 40 <= bpad <= 64).  The clouds' rounds differ in cost -- a cloud with
// more unassigned persons places more bids per round, for all 3000 rounds: +-25 % in time
// (profiles/r3_emd_cloud_times.txt) -- and a launch lasts as long as its slowest cloud.  The number
// of persons still unassigned at round 300 predicts the time the rest takes (correlation 0.96), so
// the clouds are ranked by it and dealt out to the XCDs (ranks 0..7: the heaviest cloud of each XCD,
// 8..15 the next, ...); the widths of an XCD's cloud slots follow from the loads (below), or `pattern`
// gives them, heaviest first, 4 bits each (e.g. 8,4,4,4,4,4,2,2; sum = the XCD's 4 * bpad / 8 workgroups).  A cluster lives inside
// one XCD (its members' plain stores meet in that XCD's L2): workgroup blockIdx is on XCD
// blockIdx % 8 (dispatch order; checked by the members' first gather as in the other launches).
__global__ __launch_bounds__(kEmdThreads) void emd_lean_tiers_kernel(
    int b, int bpad, int n, const float *__restrict__ xyz1, float *__restrict__ dist, int *assignment,
    float eps, int iters, char *scratch, int fast_ok, int it_stop, int which, unsigned long long pattern) {
  __shared__ LeanShared sh;
  const int lane = threadIdx.x & (kWave - 1);
  char *tail = scratch + (size_t)b * emd_scratch_per_cloud(n);
  const int x = (int)blockIdx.x & 7, q = (int)blockIdx.x >> 3;   // XCD, index inside it
  const int c = bpad >> 3;                                          // cloud slots per XCD
  // Every workgroup ranks the clouds itself, from the same 64 words: unassigned persons (0: finished), ties by index;
  // ranks 0..7 are the heaviest cloud of each XCD, 8..15 the next, ...: slot s of XCD x holds the cloud of rank 8 s + x.
  int key = -1, u = 0;
  if (lane < bpad) {
    const EmdHandover *h = emd_handover(tail, b, lane < b ? lane : 0);
    u = (lane < b && h->next_it) ? h->utot : 0;
    key = lane < b ? (u << 8) | (255 - lane) : (255 - lane) - 65536;
  }
  int rank = 0;
  for (int o = 0; o < bpad; ++o) rank += __builtin_amdgcn_readlane(key, o) > key ? 1 : 0;
  // slot s in lane s: its cloud and load
  int my_cloud = b, us = 0;
  for (int s = 0; s < c; ++s) {
    const unsigned long long hit = __ballot(lane < bpad && rank == s * 8 + x);
    const int src = hit ? (int)__builtin_ctzll(hit) : 0;
    const int uu = hit ? __builtin_amdgcn_readlane(u, src) : 0;
    if (lane == s) { my_cloud = hit ? src : b; us = uu; }
  }
  // The widths of this XCD's slots (lanes 0..7; sum = its 4 c workgroups).  pattern != 0: given (an experiment's, or
  // 8,4,..,4,2,2 for fewer than 8 slots).  Otherwise from the loads themselves: workgroups in proportion to the SQUARE of
  // the unassigned counts -- a heavy cloud stays heavy for all remaining rounds while a light one converges --, rounded
  // greedily to the widths the round loop is instantiated for (used when a slot is empty or the XCD has fewer than 8
  // slots).  With all 8 slots active the pattern is one of three, by the ratio of the heaviest to the lightest load:
  // >= 1.6 (the headline's uniform clouds: 100..222 unassigned at round 300): 8,5,4,4,3,3,3,2, the best of sixteen
  // measured (the greedy rule finds it in 7 of 8 XCDs there and a worse one in the eighth: +0.6 ms); < 1.25 (clouds of
  // equal load): 4 each -- a skewed pattern costs 13 % there; between: 6,5,4,4,4,3,3,3 (profiles/r3_emd_cloud_times.txt).
  int w = 0;
  // (c == 8, every slot active: one of three measured patterns, by the spread of the loads)
  const int u_first = __builtin_amdgcn_readlane(us, 0), u_last = __builtin_amdgcn_readlane(us, 7);
  if (pattern == 0ull && c == 8 && u_last > 0) {
    const float spread = (float)u_first / (float)u_last;
    pattern = spread < 1.25f ? 0x44444444ull : spread < 1.6f ? 0x33344456ull : 0x23334458ull;
  }
  if (pattern != 0ull) {
    w = lane < c ? (int)((pattern >> (4 * lane)) & 15ull) : 0;
  } else {
    const bool act = lane < c && us > 0;
    const float sq = act ? (float)us * (float)us : 0.f;
    float tot = sq;
#pragma unroll
    for (int o = 4; o >= 1; o >>= 1) tot += __shfl_xor(tot, o);
    const float target = tot > 0.f ? (float)(4 * c) * sq / tot : 0.f;
    w = act ? 2 : 0;
    int budget = 4 * c;
    {
      int ws = lane < 8 ? w : 0;
#pragma unroll
      for (int o = 4; o >= 1; o >>= 1) ws += __shfl_xor(ws, o);
      budget -= __builtin_amdgcn_readlane(ws, 0);
    }
    for (int guard = 0; guard < 64 && budget > 0; ++guard) {
      const int step = w == 6 ? 2 : 1;                       // 2, 3, 4, 5, 6, 8
      const bool ok = act && w < 8 && step <= budget;
      const float d = ok ? target - (float)w : -1.0e30f;
      float m = lane < 8 ? d : -1.0e30f;
#pragma unroll
      for (int o = 4; o >= 1; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o));
      m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 0));
      if (m < -1.0e29f) break;
      const unsigned long long win = __ballot(lane < 8 && ok && d == m);
      const int wl = (int)__builtin_ctzll(win);
      budget -= __builtin_amdgcn_readlane(step, wl);
      if (lane == wl) w += step;
    }
  }
  // offsets of the slots inside the XCD's workgroups, this workgroup's slot
  int off = lane < 8 ? w : 0;
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) {
    const int up = __shfl_up(off, o);
    if ((lane & 7) >= o) off += up;
  }
  off -= lane < 8 ? w : 0;   // exclusive
  const unsigned long long mine = __ballot(lane < 8 && w > 0 && q >= off && q < off + w);
  int W = 0, wg = 0, cloud = b;
  if (mine) {
    const int sl = (int)__builtin_ctzll(mine);
    W = __builtin_amdgcn_readlane(w, sl);
    wg = q - __builtin_amdgcn_readlane(off, sl);
    cloud = __builtin_amdgcn_readlane(my_cloud, sl);
  }
#define MVP_LEAN_BODY(WB) emd_lean_body<WB>(sh, cloud, wg, b, n, xyz1, dist, assignment, eps, iters, scratch, fast_ok, it_stop, which, 0)
  switch (W) {
    case 8: MVP_LEAN_BODY(8); break;
    case 6: MVP_LEAN_BODY(6); break;
    case 5: MVP_LEAN_BODY(5); break;
    case 4: MVP_LEAN_BODY(4); break;
    case 3: MVP_LEAN_BODY(3); break;
    case 2: MVP_LEAN_BODY(2); break;
    default: break;
  }
#undef MVP_LEAN_BODY
}

template <int WT, int RESN = 0>
static hipError_t emd_lean_launch_w(int b, int n, const float *xyz1, float *dist, int *assignment, float eps,
                                    int iters, char *scratch, int fast_ok, int it_stop, int which,
                                    hipStream_t stream, int u_stop = 0) {
  int bpad = WT == 1 ? b : (b + 7) / 8 * 8;
  if (WT == 1) {
    hipLaunchKernelGGL((emd_lean_kernel<1, RESN>), dim3(b), dim3(kEmdThreads), 0, stream, b, bpad, n, xyz1, dist,
                       assignment, eps, iters, scratch, 0, it_stop, which, u_stop);
    return hipSuccess;
  }
  void *args[] = {&b, &bpad, &n, &xyz1, &dist, &assignment, &eps, &iters, &scratch, &fast_ok, &it_stop, &which, &u_stop};
  return hipLaunchCooperativeKernel(reinterpret_cast<const void *>(emd_lean_kernel<WT, RESN>), dim3(WT * bpad),
                                    dim3(kEmdThreads), args, 0, stream);
}

template <int RESN>
static hipError_t emd_lean_launch_res(int b, int n, int w, const float *xyz1, float *dist, int *assignment, float eps,
                                      int iters, char *scratch, int fast_ok, hipStream_t stream, int u_stop) {
  if (w == 8) return emd_lean_launch_w<8, RESN>(b, n, xyz1, dist, assignment, eps, iters, scratch, fast_ok, iters, 1, stream, u_stop);
  if (w == 4) return emd_lean_launch_w<4, RESN>(b, n, xyz1, dist, assignment, eps, iters, scratch, fast_ok, iters, 1, stream, u_stop);
  if (w == 2) return emd_lean_launch_w<2, RESN>(b, n, xyz1, dist, assignment, eps, iters, scratch, fast_ok, iters, 1, stream, u_stop);
  return emd_lean_launch_w<1, RESN>(b, n, xyz1, dist, assignment, eps, iters, scratch, fast_ok, iters, 1, stream, u_stop);
}

// Runs the rounds the first kernel handed over.  Clouds that were not handed over exit at once.
// plan_every = 0: one launch with the cluster width `w` the first kernel ran with.  Otherwise (and
// w = 4, 33 <= b <= 64, n >= 4096, enough rounds): that width up to round `plan_round`, then launches of
// `plan_every` rounds each with TIERED widths (emd_lean_tiers_kernel; plan_widths: the widths of an
// XCD's 8 cloud slots, heaviest first, 4 bits each).
hipError_t emd_lean_launch(int b, int n, int w, const float *xyz1, float *dist, int *assignment, float eps,
                           int iters, char *scratch, int fast_ok, int plan_round, int plan_every,
                           unsigned long long plan_widths, int res_cap, hipStream_t stream) {
  if (res_cap != 0 && n <= kResMaxN) {
    // Clouds of at most kResMaxN points: the clustered rounds end as soon as at most |res_cap| persons are
    // unassigned and the rest runs LDS-resident on one workgroup per cloud (emd_resident.h) -- res_cap > 0: on
    // member 0 of the cloud's cluster at once, inside the same launch; res_cap < 0: in a launch of its own
    // (every cloud waits for the last one to get there: A/B, and the fallback if the fused kernels do not fit).
    const bool fused = res_cap > 0;
    int cap = res_cap > 0 ? res_cap : -res_cap;
    cap = cap > kResList ? kResList : cap;
    hipError_t e = hipErrorUnknown;
    if (fused) {
      e = n <= 2048 ? emd_lean_launch_res<2048>(b, n, w, xyz1, dist, assignment, eps, iters, scratch, fast_ok, stream, cap)
                    : emd_lean_launch_res<4096>(b, n, w, xyz1, dist, assignment, eps, iters, scratch, fast_ok, stream, cap);
      if (e != hipSuccess) (void)hipGetLastError();
    }
    if (e != hipSuccess) e = emd_lean_launch_res<0>(b, n, w, xyz1, dist, assignment, eps, iters, scratch, fast_ok, stream, cap);
    if (e != hipSuccess) return e;
    // (fused: only clouds nobody finished -- none -- are left for it; it exits at once)
    return emd_resident_launch(b, n, dist, assignment, eps, iters, scratch, stream);
  }
  int bpad = (b + 7) / 8 * 8;
  const int c = bpad / 8;
  // widths of an XCD's cloud slots: 0 = from the loads (the kernel), else the caller's, heaviest first, 4 bits each
  unsigned long long pattern = c == 8 ? plan_widths : 0ull;
  bool ok = true;
  if (pattern != 0ull) {
    int sum = 0;
    for (int s = 0; s < c; ++s) {
      const int ws = (int)((pattern >> (4 * s)) & 15ull);
      ok = ok && (ws == 2 || ws == 3 || ws == 4 || ws == 5 || ws == 6 || ws == 8);
      sum += ws;
    }
    ok = ok && sum == 4 * c;
  }
  // (below 4096 points a cloud has a few dozen bidders left at round 300: nothing to deal out -- measured: 2048 points +0.7 ms)
  if (w == 4 && n >= 4096 && b >= 33 && b <= 64 && plan_every > 0 && ok && iters >= plan_round + 256) {
    hipError_t e = emd_lean_launch_w<4>(b, n, xyz1, dist, assignment, eps, iters, scratch, fast_ok, plan_round, 1, stream);
    for (int r = plan_round; e == hipSuccess && r < iters; r += plan_every) {
      int stop = r + plan_every + 256 > iters ? iters : r + plan_every;   // (no short last launch)
      int which = 2;
      void *args[] = {&b, &bpad, &n, &xyz1, &dist, &assignment, &eps, &iters, &scratch, &fast_ok, &stop, &which, &pattern};
#ifdef MVP_TEST_HOOKS
      static const bool refuse = std::getenv("MVP_EMD_TIERS_FAIL") != nullptr;   // (libmvpops_hooks.so only: behave as if it did not fit)
#else
      constexpr bool refuse = false;
#endif
      e = refuse ? hipErrorCooperativeLaunchTooLarge
                 : hipLaunchCooperativeKernel(reinterpret_cast<const void *>(emd_lean_tiers_kernel), dim3(4 * bpad),
                                              dim3(kEmdThreads), args, 0, stream);
      if (e != hipSuccess) {
        // (the tiered kernel does not fit this device: the fixed-width kernel finishes what the first launch left)
        (void)hipGetLastError();
        return emd_lean_launch_w<4>(b, n, xyz1, dist, assignment, eps, iters, scratch, fast_ok, iters, 2, stream);
      }
      if (stop == iters) break;
    }
    return e;
  }
  if (w == 8) return emd_lean_launch_w<8>(b, n, xyz1, dist, assignment, eps, iters, scratch, fast_ok, iters, 1, stream);
  if (w == 4) return emd_lean_launch_w<4>(b, n, xyz1, dist, assignment, eps, iters, scratch, fast_ok, iters, 1, stream);
  if (w == 2) return emd_lean_launch_w<2>(b, n, xyz1, dist, assignment, eps, iters, scratch, fast_ok, iters, 1, stream);
  return emd_lean_launch_w<1>(b, n, xyz1, dist, assignment, eps, iters, scratch, fast_ok, iters, 1, stream);
}

}  // namespace mvp
