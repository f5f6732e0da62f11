// Auction-based Earth Mover's Distance for gfx950.
//
// Replaces emd_cuda_forward / emd_cuda_backward and their 9 kernels
// (utils/metrics/EMD/emd_cuda.cu:23-226, 228-282, 284-316).
//
// The reference runs 7 kernel launches per auction round on the legacy
// default stream (21 001 launches at the eval setting iters=3000), every
// bidder scans all n objects every round (O(n^2 k)), and GetMax is racy.
// MI355X-first design:
//   * ONE persistent launch.  A cloud is owned by a CLUSTER of W 1024-lane
//     workgroups (W = 1, 2, 4 or 8, chosen so that b*W workgroups fit the chip's
//     CUs: at the headline batch of 64 clouds a single workgroup per cloud
//     would leave 192 of 256 CUs idle).  Every workgroup keeps its own
//     unassigned list and bids for it; the auction state (prices, owners,
//     bid keys) is shared through write-through (sc1) stores and L1-bypassing
//     (sc1) loads, and the round structure is kept by two all-gathers of
//     8-byte tagged granules per round (the data is the flag; no fences in
//     the loop).  When the members find themselves on one XCD (the launch
//     places them so, HW_REG_XCC_ID tells) they share its L2 and the stores
//     stay plain.  W = 1 uses plain accesses and workgroup barriers only;
//   * rounds stop as soon as nobody is unassigned (exact: such rounds are
//     no-ops in the reference, emd_cuda.cu:105-106,185,199);
//   * the unassigned lists are maintained incrementally (losers stay, an
//     evicted owner joins the list of the workgroup that evicted it) instead
//     of a count / prefix-sum / compaction pass over all n points per round;
//   * objects (xyz2) are sorted once along a Hilbert curve and stored in that
//     order as float4 {x, y, z, price}; 16 consecutive slots are a LEAF, 16
//     consecutive leaves a NODE (emd_index.h: the index follows the cloud --
//     16 objects per leaf on a volume and on a 2-manifold alike; rounds 1-5 used
//     a uniform 12^3 grid).  Per leaf and node every workgroup keeps in LDS the
//     exact bounding box of the members and a lower bound of their prices
//     (prices only rise, so a stale bound stays valid; refreshed bounds are
//     broadcast to the other workgroups of the cluster each round);
//   * Bid.  A bid needs the best and second-best of
//         v_k = float(3.0 - (double)sqrtf(|q-o_k|^2) - price_k)     (emd_cuda.cu:146)
//     over all k.  Instead of evaluating all n objects the bidder (1) seeds a
//     lower bound B2 of the second-best value from its home chunk and the
//     chunks of its previous two best objects, (2) tests the <= 64 nodes and
//     the leaves of the passing nodes against
//     dist(q, box) + price_lb <= 3 - B2  and
//     (3) visits only surviving leaves, where each object first passes the same
//     conservative test on its squared distance (no sqrt, no double); only
//     objects that can still change {best, second best, best index} get the
//     exact double-precision value.  Every skip is provably lossless (kMargin
//     below), so bids are bit-identical to the exhaustive scan.  Two
//     schedules share that logic: rounds with many bidders run FOUR bidders
//     per wave (one per 16-lane DPP row; lane-local exact top-2, merged per
//     row with DPP) for throughput; rounds with few bidders run one bidder
//     per wave (wave-uniform state, scalar folds; emd_search_wave.inc) for the
//     shortest dependent chain.  A search that finds the whole cloud within
//     reach falls back to a linear scan of the sorted objects;
//   * ties are resolved by the reference's own order, reconstructed from its
//     thread partition (emd_cuda.cu:108-118,139-142,163-171): candidates are
//     ordered by (chunk thread, 2048-tile, index in tile) of their ORIGINAL
//     object index;
//   * GetMax (emd_cuda.cu:181-194) is folded into the bid: a bid is ONE 64-bit
//     atomic max of {order-preserving bits of the increment, bidder} on the
//     object, so after the round's barrier the key already names the winner
//     -- the highest bidder index among the maximal increments, which is what
//     executing the reference's racy kernel sequentially gives.  The
//     reference lets every bidder within 1e-6 of the maximum compete
//     (emd_cuda.cu:188); the atomic's return value tells a bidder whether a
//     different increment within that band was bid on the same object, and
//     only rounds where that happened (a few per 10^5 bids) run the explicit
//     GetMax pass and its extra barrier.
#include <mutex>

#include "emd_common.h"
#include "emd_index.h"

namespace mvp {

// emd_lean.hip: the kernel that runs the one-bidder-per-wave rounds after the hand-over
hipError_t emd_lean_launch(int b, int n, int w, const float *xyz1, float *dist, int *assignment, float eps,
                           int iters, char *scratch, int fast_ok, int plan_round, int plan_every,
                           unsigned long long plan_widths, int res_cap, hipStream_t stream);

template <int W>
__global__ __launch_bounds__(kEmdThreads) void emd_auction_kernel(
    int b, int bpad, int n, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
    float *__restrict__ dist, int *assignment, float eps, int iters, char *scratch, int fast_ok, int lean) {
  // Block -> (cloud, member): members of a cluster are bpad (a multiple of 8)
  // blocks apart, so they share an XCD under the round-robin dispatch (faster
  // L2 sharing only; nothing below depends on placement).
  const int cloud = W == 1 ? (int)blockIdx.x : (int)blockIdx.x % bpad;
  const int wg = W == 1 ? 0 : (int)blockIdx.x / bpad;
  if (cloud >= b) return;
  char *cbase = scratch + (size_t)cloud * emd_scratch_per_cloud(n);
  char *tail = scratch + (size_t)b * emd_scratch_per_cloud(n);
#ifdef MVP_EMD_CLOUDTIME
  const long long ct_a0 = wall_clock64();
#endif
  u64 *slots = emd_granules(tail, b, cloud, 0);
  // per-cloud auction statistics {rounds executed, bids made} (read by
  // bench.py; not part of the op's result)
  EmdHandover *resume = emd_handover(tail, b, cloud);
  long long *stats = emd_stats(tail, b, cloud);
  const int t = threadIdx.x;
  const int lane = t & (kWave - 1);
  const int wave = t >> 6;
  xyz1 += (size_t)cloud * n * 3;
  xyz2 += (size_t)cloud * n * 3;
  dist += (size_t)cloud * n;
  int *ass = assignment + (size_t)cloud * n;
  const EmdScratch sc = emd_carve(cbase, n);

  // ---- accessors of the shared auction state.  W == 1: plain.  W > 1: loads
  // bypass this CU's L1 (sc1), stores are written through (sc1), so data is
  // visible to the other workgroups once the store is acknowledged.
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(cbase, 0, (int)emd_scratch_per_cloud(n), 0x00020000);
  auto ld_obj = [&](int s) -> float4 {
    if constexpr (W == 1) {
      return sc.obj[s];
    } else {
      const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)s * 16u, 0, 16);
      return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
  };
  auto ld_price = [&](int s) -> float {  // obj[s].w alone
    if constexpr (W == 1) return sc.obj[s].w;
    else return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)s * 16u + 12u, 0, 16));
  };
  auto ld_ostate = [&](int s) -> int4 {
    if constexpr (W == 1) {
      return sc.ostate[s];
    } else {
      const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, ((unsigned)n + (unsigned)s) * 16u, 0, 16);
      return make_int4((int)v.x, (int)v.y, (int)v.z, (int)v.w);
    }
  };
  auto ld_person = [&](int j, int half) -> float4 {  // half 0 = lo, 1 = hi
    if constexpr (W == 1) {
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
    if constexpr (W == 1) {
      sc.person[2 * j + 1] = make_float4(__int_as_float(bid), __int_as_float(p1), __int_as_float(p2), inc);
    } else {
      v4u v;
      v.x = (unsigned)bid; v.y = (unsigned)p1; v.z = (unsigned)p2; v.w = __float_as_uint(inc);
      if (same_xcd) __builtin_amdgcn_raw_buffer_store_b128(v, rs, (2u * (unsigned)n + 2u * (unsigned)j + 1u) * 16u, 0, 0);
      else __builtin_amdgcn_raw_buffer_store_b128(v, rs, (2u * (unsigned)n + 2u * (unsigned)j + 1u) * 16u, 0, 16);
    }
  };
  auto st_ostate = [&](int s, int owner) {  // key = 0 (no bid), new owner
    if constexpr (W == 1) {
      sc.ostate[s] = make_int4(0, 0, owner, 0);
    } else {
      v4u v;
      v.x = 0u; v.y = 0u; v.z = (unsigned)owner; v.w = 0u;
      if (same_xcd) __builtin_amdgcn_raw_buffer_store_b128(v, rs, ((unsigned)n + (unsigned)s) * 16u, 0, 0);
      else __builtin_amdgcn_raw_buffer_store_b128(v, rs, ((unsigned)n + (unsigned)s) * 16u, 0, 16);
    }
  };
  auto st_i32 = [&](int *p, int v) {
    if constexpr (W == 1) *p = v;
    else if (same_xcd) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto st_f32 = [&](float *p, float v) {
    if constexpr (W == 1) *p = v;
    else if (same_xcd) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto ld_key = [&](int s) -> u64 {
    u64 *p = reinterpret_cast<u64 *>(&sc.ostate[s]);
    if constexpr (W == 1) return *p;
    else return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  // The index (emd_index.h): per leaf and per node {box min x, y, z, price lower bound} and {box max x, y, z, -}:
  // a box test is two ds_read_b128 per lane.  (During the build the leaves' 32 KB hold the sort's 512 x 16 counters.)
  __shared__ float4 l_box[2 * kMaxLeaves];
  float4 *const l_lo = l_box, *const l_hi = l_box + kMaxLeaves;
  __shared__ float4 n_lo[kMaxNodes], n_hi[kMaxNodes];
  __shared__ int s_leafkey[kMaxLeaves];  // build only: first Hilbert key of every leaf (the persons' home chunks)
  __shared__ unsigned short w_list[kEmdWaves][4 * kRowListCap];  // surviving leaves, per 16-lane row
  __shared__ float s_red[6][kEmdWaves];
  __shared__ int s_wsum[kEmdWaves];
  __shared__ int s_cnt[2];
  __shared__ int s_next;             // next undrawn list position of the round (wave-mode bids)
  __shared__ int s_err, s_abort, s_nchg, s_xcc;
  __shared__ int s_alarm[2];  // by round parity: set in Bid, read after the barrier, cleared a round later
  __shared__ unsigned s_gout[2 * kMaxCluster];
#ifdef MVP_EMD_PROFILE
  __shared__ int s_wbusy[kEmdWaves];
  __shared__ unsigned long long s_hist2[4];
  if (threadIdx.x < 4) s_hist2[threadIdx.x] = 0;
  __shared__ unsigned long long s_slow[2][8];  // [d >= 10k cycles][count, nsub, cells, visit steps, extra member iterations, folds, seed cycles, visit cycles]
  if (threadIdx.x < 16) s_slow[threadIdx.x >> 3][threadIdx.x & 7] = 0;
  __shared__ unsigned long long s_hist[16];  // wave-mode bids: [0..7] duration buckets, [8] sum nsub, [9] sum cells visited, [10] count, [11] linear scans, [12] sum cycles
  if (threadIdx.x < 16) s_hist[threadIdx.x] = 0;
#endif
  // this round's bids for the first kBidCache list positions (skips two
  // dependent global round trips in Assign)
  __shared__ int s_bj[kBidCache], s_bo[kBidCache], s_b2k[kBidCache];
  __shared__ float s_binc[kBidCache];
  // person records {qx,qy,qz,-} / {j, prev1, prev2, -} of the first kRecCap
  // entries of the current / next unassigned list: in the long tail of the
  // auction a bid starts from LDS instead of two dependent global reads
  __shared__ float4 s_rq[2][kRecCap];
  __shared__ int4 s_ri[2][kRecCap];

  // ------------------------------------------------------------ index build (emd_index.h)
  // Member 0 alone sorts the objects along the Hilbert curve and writes the shared arrays and the initial state;
  // after the cluster's barrier every member derives the leaves' and nodes' boxes from the sorted objects itself.
  const int lshift = emd_leaf_shift(n), nleaf = n >> lshift, nnode = (nleaf + kNodeFan - 1) / kNodeFan;
  const int kch = 1 << (lshift - 4);   // 16-slot chunks per leaf (1 up to 16384 points)
  if (t == 0) {
    s_err = 0;
    s_abort = 0;
    s_alarm[0] = 0;
    s_alarm[1] = 0;
    s_next = kEmdWaves;
    s_nchg = 0;
  }
  if (wg == 0) {
    // (a) the cube that bounds both clouds
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int k = t; k < n; k += kEmdThreads) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float u = xyz1[k * 3 + a], w = xyz2[k * 3 + a];
        mn[a] = __builtin_fminf(mn[a], __builtin_fminf(u, w));
        mx[a] = __builtin_fmaxf(mx[a], __builtin_fmaxf(u, w));
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        mn[a] = __builtin_fminf(mn[a], __shfl_xor(mn[a], off, kWave));
        mx[a] = __builtin_fmaxf(mx[a], __shfl_xor(mx[a], off, kWave));
      }
      if (lane == 0) {
        s_red[a][wave] = mn[a];
        s_red[3 + a][wave] = mx[a];
      }
    }
    __syncthreads();
    HilbertFrame hf;
    {
      float lo[3], hi[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        lo[a] = s_red[a][0];
        hi[a] = s_red[3 + a][0];
        for (int w = 1; w < kEmdWaves; ++w) {
          lo[a] = __builtin_fminf(lo[a], s_red[a][w]);
          hi[a] = __builtin_fmaxf(hi[a], s_red[3 + a][w]);
        }
      }
      float ext = __builtin_fmaxf(hi[0] - lo[0], __builtin_fmaxf(hi[1] - lo[1], hi[2] - lo[2]));
      if (!(ext > 0.f) || !(ext < 3.0e38f)) ext = 1.f;
      hf.lox = lo[0];
      hf.loy = lo[1];
      hf.loz = lo[2];
      hf.scale = (float)(1 << kHilbertBits) / ext;
    }
    // (b) {Hilbert key, object} pairs, sorted by three stable 9-bit passes through the (not yet used) list areas
    u64 *buf_a = reinterpret_cast<u64 *>(sc.ulist), *buf_b = buf_a + n;   // 16 n of the lists' 64 n bytes
    for (int k = t; k < n; k += kEmdThreads)
      buf_a[k] = ((u64)emd_hilbert_key(hf, xyz2[k * 3 + 0], xyz2[k * 3 + 1], xyz2[k * 3 + 2]) << 32) | (u64)(unsigned)k;
    __syncthreads();
    int *hist = reinterpret_cast<int *>(l_box);
    static_assert(sizeof(float4) * 2 * kMaxLeaves >= sizeof(int) * kSortDigits * kEmdWaves, "the counters fit the leaves' LDS");
    emd_sort_pass(buf_a, buf_b, n, 0, hist, s_wsum);
    emd_sort_pass(buf_b, buf_a, n, kHilbertBits, hist, s_wsum);
    emd_sort_pass(buf_a, buf_b, n, 2 * kHilbertBits, hist, s_wsum);
    // (c) the sorted objects; initial state of emd_module.py:54-65
    for (int s = t; s < n; s += kEmdThreads) {
      const u64 e = buf_b[s];
      const int k = (int)(unsigned)e;
      sc.obj[s] = make_float4(xyz2[k * 3 + 0], xyz2[k * 3 + 1], xyz2[k * 3 + 2], 0.f);
      sc.perm[s] = k;
      ass[s] = -1;
      sc.ostate[s] = make_int4(0, 0, -1, 0);
      if ((s & ((1 << lshift) - 1)) == 0) s_leafkey[s >> lshift] = (int)(unsigned)(e >> 32);
    }
    __syncthreads();
    // (d) the persons' records: point + home chunk (the leaf whose key range holds the person's own key)
    for (int k = t; k < n; k += kEmdThreads) {
      const float x = xyz1[k * 3 + 0], y = xyz1[k * 3 + 1], z = xyz1[k * 3 + 2];
      const int key = (int)emd_hilbert_key(hf, x, y, z);
      int lo = 0, hi = nleaf;   // last leaf whose first key is <= key (leaf 0 if none)
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_leafkey[mid] <= key) lo = mid; else hi = mid;
      }
      sc.person[2 * k] = make_float4(x, y, z, __int_as_float(lo << (lshift - 4)));
      sc.person[2 * k + 1] = make_float4(__int_as_float(-1), __int_as_float(-1), __int_as_float(-1), 0.f);
    }
  }
  int share = n / W;  // n % 1024 == 0
  int first = wg * share;
  int *my_ulist = sc.ulist + (size_t)wg * 2 * n;
  unsigned epoch = 0;
  if constexpr (W > 1) {
    // hand the shared arrays to the other members: release (write back this
    // XCD's L2) -> barrier -> acquire (drop this CU's L1) -> plain loads
    __syncthreads();
    if (wg == 0 && t == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (t == 0) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      s_xcc = (int)(xcc & 0xFu) + 1;
    }
    const bool ok = emd_cluster_gather<W>(slots, wg, ++epoch, &s_xcc, &s_xcc, s_gout, &s_abort);
    if (t == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    // every member on one XCD (the launch places them so; the registers say
    // whether it happened): the cheaper same-L2 store flavour is valid
    same_xcd = fast_ok != 0;
#pragma unroll
    for (int w = 1; w < W; ++w) same_xcd &= s_gout[2 * w] == s_gout[0];
    if (!ok) {
      if (wg == 0 && t == 0) stats[0] = -2;
      for (int j = t; j < n; j += kEmdThreads) {
        dist[j] = __builtin_nanf("");
        ass[j] = -1;
      }
      return;
    }
  } else {
    __syncthreads();
  }
  // (e) every member: exact box per leaf and node, price lower bounds 0 (emd_index.h; ends with a barrier) ...
  emd_index_boxes(l_lo, l_hi, n_lo, n_hi, sc.obj, n, lshift);
  // ... and its own share of the persons as its first unassigned list (the list areas were the sort's buffers)
  for (int k = t; k < share; k += kEmdThreads) my_ulist[k] = first + k;
  if (t < kRecCap && t < share) {  // round 0: list position u holds person first + u
    const int k = first + t;
    s_rq[0][t] = sc.person[2 * k];
    s_ri[0][t] = make_int4(k, -1, -1, 0);
  }
  if (t == 0) {
    s_cnt[0] = share;
    s_cnt[1] = 0;
  }
  __syncthreads();

  // ------------------------------------------------------------ the auction
  const int block_cnt = n / 1024;
  int cur = 0;
  int Utot = n;  // unassigned persons of the whole cloud
  long long n_rounds = 0, n_bids = 0;
  bool aborted = false;
  u64 chg_pend = 0ull;    // a bound broadcast by another workgroup, not yet folded in
  bool chg_have = false;
  // Once at most kSoloMax persons are left (their number never grows) one
  // workgroup bids for all of them in a single pass and the cluster's barriers
  // would only add latency: member 0 adopts the others' lists and carries on
  // alone (same code, workgroup barriers), the others leave.
  bool clustered = W > 1;
#ifdef MVP_EMD_PROFILE
  long long prof_pg1 = 0, prof_drain = 0, prof_gather = 0, cyc_bid = 0, cyc_sync1 = 0, cyc_assign = 0, cyc_sync2 = 0, n_alarm = 0, n_rebal = 0, prof_u = 0, prof_a1 = 0, prof_an = 0, prof_a2 = 0, prof_a3 = 0, prof_a4 = 0;
#endif
#ifdef MVP_EMD_PROFILE
  const long long t_loop0 = __builtin_readcyclecounter();
#endif
  for (int it = 0; it < iters; ++it) {
    if (Utot == 0) break;
#ifdef MVP_EMD_PROFILE
    if (cloud == 0 && wg == 0 && t == 0 && (it == 1 || it == 2 || it == 3 || it == 5 || it == 10 || it == 25 || it == 50 || it == 100 || it == 150 || it == 250 || it == 500 || it == 750 ||
                                           it == 1000 || it == 1500 || it == 2000 || it == 2500 || it == iters - 1))
      printf("head cloud 0: round %d starts at %lld cycles, unassigned %d\n", it, __builtin_readcyclecounter() - t_loop0, Utot);
#endif
    const int U = s_cnt[cur];  // this workgroup's bidders
    n_rounds += 1;
    n_bids += U;
    const bool last = it == iters - 1;
    const int *L = my_ulist + (size_t)cur * n;
    int *Lnext = my_ulist + (size_t)(cur ^ 1) * n;
    // thread_per_unass of the reference (emd_cuda.cu:107-109): fixes the tie
    // order only.
    const int upb = (Utot + block_cnt - 1) / block_cnt;
    const int tpu = 1024 / upb;

#ifdef MVP_EMD_PROFILE
    const long long tp0 = __builtin_readcyclecounter();
#endif
    // A bid = one returning 64-bit atomic max on the object's key; the value
    // it returns is examined one bid later (or after the loop), so the wave
    // never waits for it.
    u64 pend_old = 0ull;
    float pend_inc = 0.f;
    bool alarm = false;
    // Two Bid paths: many bidders -> four bidders per wave (throughput);
    // few bidders -> one bidder per wave (shortest dependent chain).
    if (U > kRowModeMin) {
    // ---------------- Bid (emd_cuda.cu:95-179): one 16-lane ROW per bidder
    // A bid touches ~10 leaves and ~160 objects, so a 64-lane wave per bidder
    // mostly waits on its own dependent instruction stream.  Four bidders per
    // wave (one per 16-lane DPP row) run those streams side by side: row-wide
    // reductions are 4 DPP steps, every lane keeps the exact top-2 of the
    // candidates it evaluated, and the rows are merged once per bid.
    {
      const int row = lane >> 4, l16 = lane & 15;
      const int rsh = row * 16;
      unsigned short *wl = w_list[wave] + row * kRowListCap;
      for (int ub = 0; ub < U; ub += kEmdWaves * 4) {
        const int u = ub + wave * 4 + row;
        const bool act = u < U;  // row-uniform
        float4 ra = make_float4(0.f, 0.f, 0.f, 0.f);
        int4 rb = make_int4(0, -1, -1, 0);
        if (act) {
          if (u < kRecCap) {
            ra = s_rq[cur][u];
            rb = s_ri[cur][u];
          } else {
            const int jj = L[u];
            ra = ld_person(jj, 0);
            const float4 g = ld_person(jj, 1);
            rb = make_int4(jj, __float_as_int(g.y), __float_as_int(g.z), 0);
          }
        }
        const int j = rb.x;
        const float qx = ra.x, qy = ra.y, qz = ra.z;
        const int p1 = rb.y, p2 = rb.z;
        const int hc = act ? __float_as_int(ra.w) : 0;   // home chunk (emd_index.h)

        // (1) seed: second-largest exact value among DISTINCT real objects -- the
        // home chunk's 16 members, the previous best / second best when they live
        // elsewhere and, before the first bid, the home chunk's sibling (a valid
        // lower bound of the final second best).
        float tm;
        {
          float a1 = -1e9f, a2 = -1e9f;
          // (the hint objects' load is issued first: it shares the round trip of the home chunk's)
          const bool hint = act && ((l16 == 0 && p1 >= 0 && (p1 >> 4) != hc) || (l16 == 1 && p2 >= 0 && (p2 >> 4) != hc));
          const bool two = act && p1 < 0 && p2 < 0;   // row-uniform
          float4 oh = make_float4(0.f, 0.f, 0.f, 0.f), o0 = oh, o1 = oh;
          if (hint) oh = ld_obj(l16 == 0 ? p1 : p2);
          if (act) o0 = ld_obj(hc * 16 + l16);
          if (two) o1 = ld_obj((hc ^ 1) * 16 + l16);
          if (act) top2_insert(a1, a2, emd_value(sqdist3(o0.x - qx, o0.y - qy, o0.z - qz), o0.w));
          if (two) top2_insert(a1, a2, emd_value(sqdist3(o1.x - qx, o1.y - qy, o1.z - qz), o1.w));
          if (hint) top2_insert(a1, a2, emd_value(sqdist3(oh.x - qx, oh.y - qy, oh.z - qz), oh.w));
          top2_dpp_step<0xB1, 0xF>(a1, a2);   // butterfly inside the row:
          top2_dpp_step<0x4E, 0xF>(a1, a2);   // every lane ends with the row's
          top2_dpp_step<0x141, 0xF>(a1, a2);  // (largest, second largest)
          top2_dpp_step<0x140, 0xF>(a1, a2);
          tm = (3.0f - a2) + kMargin;
        }

        // exact top-2 of the candidates THIS LANE evaluates
        float lb1 = -1e9f, lb2 = -1e9f;
        int lbk = -1, lb2k = -1;
        auto consider = [&](bool in_range, const float4 &o, int slot) {
          const float sd = sqdist3(o.x - qx, o.y - qy, o.z - qz);
          const float tq = tm - o.w;
          const bool ps = in_range && tq >= 0.f && sd <= tq * tq;
          if (__any(ps)) {
            // (selects, not branches: with branches the compiler keeps the two slots in a
            // dynamically indexed stack array -- scratch traffic in this loop)
            const float v = ps ? emd_value(sd, o.w) : -2e9f;
            const bool gt = v > lb1, eq = ps && v == lb1;
            bool first = false;   // rare tie for the best: reference order on ORIGINAL indices
            if (__any(eq)) {
              if (eq) first = emd_precedes(sc.perm[slot], sc.perm[lbk], n, tpu);
            }
            const bool g2 = v > lb2;
            const int nk2 = gt ? lbk : (eq ? (first ? lbk : slot) : (g2 ? slot : lb2k));
            const float n2 = gt ? lb1 : (eq ? v : (g2 ? v : lb2));
            lbk = (gt || first) ? slot : lbk;
            lb1 = gt ? v : lb1;
            lb2k = nk2;
            lb2 = n2;
          }
        };

        // (2) the nodes, 16 per step: this row's passing nodes as a bit mask
        unsigned long long nm = 0ull;
        for (int nb = 0; nb < nnode; nb += 16) {
          const bool np = act && emd_box_pass(n_lo[nb + l16], n_hi[nb + l16], qx, qy, qz, tm);
          nm |= ((__ballot(np) >> rsh) & 0xFFFFull) << nb;
        }
        // the whole cloud within reach (clustered prediction against a spread target): every node passes AND so do
        // all 16 leaves of the row's first step -> the sorted objects are scanned linearly instead (below)
        bool all_near = act && __builtin_popcountll(nm) >= nnode;
        bool linear = false;

        // (4) visit the listed leaves of this row, kRowVisitLoads per step (16 lanes each)
        int nlist = 0;
        auto visit = [&]() {
          for (int k0 = 0; __any(k0 < nlist); k0 += kRowVisitLoads) {
            int s[kRowVisitLoads];
            bool in[kRowVisitLoads];
#pragma unroll
            for (int r4 = 0; r4 < kRowVisitLoads; ++r4) {
              in[r4] = k0 + r4 < nlist;
              s[r4] = in[r4] ? ((int)wl[k0 + r4] << lshift) + l16 : 0;
            }
            for (int c = 0; c < kch; ++c) {   // (one chunk per leaf up to 16384 points)
              float4 o[kRowVisitLoads];
#pragma unroll
              for (int r4 = 0; r4 < kRowVisitLoads; ++r4)
                o[r4] = in[r4] ? ld_obj(s[r4] + 16 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
              for (int r4 = 0; r4 < kRowVisitLoads; ++r4) consider(in[r4], o[r4], s[r4] + 16 * c);
            }
          }
          nlist = 0;
        };
        // (3) the 16 leaves of every passing node, one node of this row per step
        while (__any(nm != 0ull)) {
          bool lp = false;
          int leaf = 0;
          if (nm != 0ull) {
            leaf = (int)__builtin_ctzll(nm) * kNodeFan + l16;
            nm &= nm - 1ull;
            lp = emd_box_pass(l_lo[leaf], l_hi[leaf], qx, qy, qz, tm);
          }
          const unsigned rmask = (unsigned)((__ballot(lp) >> rsh) & 0xFFFFull);
          if (__builtin_expect(all_near && rmask == 0xFFFFu, 0)) {   // (row-uniform; only ever in the first step)
            linear = true;
            nm = 0ull;
            lp = false;
          }
          all_near = false;
          const unsigned amask = linear ? 0u : rmask;
          if (lp) wl[nlist + __builtin_popcount(amask & ((1u << l16) - 1u))] = (unsigned short)leaf;
          nlist += __builtin_popcount(amask);
          if (__any(nlist > kRowListCap - 16)) visit();  // keep room for the next node's 16 leaves
        }
        visit();
        if (__builtin_expect(__any(linear), 0)) {
          for (int base = 0; base < n; base += 64) {   // n % 1024 == 0
            float4 o[4];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
              o[r4] = linear ? ld_obj(base + r4 * 16 + l16) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) consider(linear, o[r4], base + r4 * 16 + l16);
          }
        }

        // (4) merge the 16 lanes of the row: exact best / second best with the
        // reference's tie order (original object indices ride along)
        int lo1 = lbk >= 0 ? sc.perm[lbk] : 0;
        auto merge_step = [&](auto ctrl_tag) {
          constexpr int CTRL = decltype(ctrl_tag)::value;
          const float ob1 = dpp_f32<CTRL, 0xF>(-1e9f, lb1);
          const float ob2 = dpp_f32<CTRL, 0xF>(-1e9f, lb2);
          const int obk = __builtin_amdgcn_update_dpp(-1, lbk, CTRL, 0xF, 0xF, false);
          const int ob2k = __builtin_amdgcn_update_dpp(-1, lb2k, CTRL, 0xF, 0xF, false);
          const int oo1 = __builtin_amdgcn_update_dpp(0, lo1, CTRL, 0xF, 0xF, false);
          const bool tie = ob1 == lb1 && obk >= 0 && lbk >= 0;
          bool other_first = false;
          if (__builtin_expect(__any(tie), 0)) other_first = tie && emd_precedes(oo1, lo1, n, tpu);
          const bool other_wins = ob1 > lb1 || other_first;
          if (other_wins) {
            const bool from_b1 = lb1 >= ob2;
            lb2 = from_b1 ? lb1 : ob2;
            lb2k = from_b1 ? lbk : ob2k;
            lb1 = ob1; lbk = obk; lo1 = oo1;
          } else {
            const bool from_ob1 = ob1 >= lb2;
            lb2k = from_ob1 ? obk : lb2k;
            lb2 = from_ob1 ? ob1 : lb2;
          }
        };
        merge_step(std::integral_constant<int, 0xB1>{});
        merge_step(std::integral_constant<int, 0x4E>{});
        merge_step(std::integral_constant<int, 0x141>{});
        merge_step(std::integral_constant<int, 0x140>{});

        if (act && lbk < 0) {  // cannot happen (>= 2 objects always survive); never index with -1
          if (l16 == 0) s_err = 1;
          lbk = 0;
          lb2k = -1;
        }
        if (act && l16 == 0) {
          const float inc = lb1 - lb2 + eps;
          st_person_hi(j, lbk, lbk, lb2k, inc);
          if (u < kBidCache) {
            s_bj[u] = j;
            s_bo[u] = lbk;
            s_b2k[u] = lb2k;
            s_binc[u] = inc;
          }
          alarm |= emd_band_alarm(pend_old, pend_inc);
          pend_old = atomicMax(reinterpret_cast<u64 *>(&sc.ostate[lbk]),
                               ((u64)emd_f2ord(inc) << 32) | (u64)((unsigned)j + 1u));
          pend_inc = inc;
        }
      }
    }
    } else {
    // ---------------- Bid (emd_cuda.cu:95-179): one wave per bidder
    // Only rounds with at most kRowModeMin (< kRecCap) bidders come here, so
    // every bidder's record is in the LDS cache.  Bids differ in length (4k to
    // 16k cycles), so a wave that finishes draws the next list position from a
    // shared counter instead of owning every 16th one: the phase lasts
    // sum / 16 instead of the longest pair.  (The loop is bounded independently
    // of the drawn position.)
    static_assert(kRowModeMin < kRecCap, "wave-mode bidders must have their records in LDS");
    int u = wave;
    for (int guard = 0; guard <= kRowModeMin && u < U; ++guard) {
      const float4 ra = s_rq[cur][u];
      const int4 rb = s_ri[cur][u];
      int drawn = 0;
      const int j = rb.x;
      const float qx = ra.x, qy = ra.y, qz = ra.z;
      const int p1 = rb.y, p2 = rb.z;
#ifdef MVP_EMD_PROFILE
      const long long tb0 = __builtin_readcyclecounter();
      int prof_cells = 0;
#endif
      const int hc = __float_as_int(ra.w);   // home chunk (emd_index.h)
      BidState st;
      st.b1 = -1e9f;
      st.b2 = -1e9f;
      st.bk = -1;
      st.b2k = -1;
      unsigned short *wl = w_list[wave];
      int nsub = 0;          // leaves tested (statistics)
#ifdef MVP_EMD_PROFILE
      long long tb1 = tb0;
      float prof_tm_seed = 0.f;
      long long t_visit = 0;
      int n_visit = 0, prof_fold = 0, prof_more = 0;
#endif
#define EMD_SEARCH_FOLD(m_, v_, slot_, price_) emd_fold(st, m_, v_, slot_, n, tpu, sc.perm)
#include "emd_search_wave.inc"
#undef EMD_SEARCH_FOLD
      (void)nsub;
#ifdef MVP_EMD_PROFILE
      if (lane == 0 && it >= 100) {
        const long long d = __builtin_readcyclecounter() - tb0;
        int bkt = 0;
        while (bkt < 7 && d >= (2000ll << bkt)) ++bkt;
        atomicAdd(&s_hist[bkt], 1ull);
        atomicAdd(&s_hist[8], (unsigned long long)nsub);
        atomicAdd(&s_hist[9], (unsigned long long)prof_cells);
        atomicAdd(&s_hist[10], 1ull);
        if (linear) atomicAdd(&s_hist[11], 1ull);
        atomicAdd(&s_hist[12], (unsigned long long)d);
        atomicAdd(&s_hist2[0], (unsigned long long)(tb1 - tb0));
        atomicAdd(&s_hist2[1], (unsigned long long)t_visit);
        atomicAdd(&s_hist2[2], (unsigned long long)n_visit);
        atomicAdd(&s_hist2[3], (unsigned long long)prof_fold);
        unsigned long long *sl = s_slow[d >= 10000 ? 1 : 0];
        atomicAdd(&sl[0], 1ull);
        atomicAdd(&sl[1], (unsigned long long)nsub);
        atomicAdd(&sl[2], (unsigned long long)prof_cells);
        atomicAdd(&sl[3], (unsigned long long)n_visit);
        atomicAdd(&sl[4], (unsigned long long)prof_more);
        atomicAdd(&sl[5], (unsigned long long)prof_fold);
        atomicAdd(&sl[6], (unsigned long long)(tb1 - tb0));
        atomicAdd(&sl[7], (unsigned long long)t_visit);
      }
#endif
      if (st.bk < 0) {  // cannot happen (>= 2 objects always survive); never index with -1
        if (lane == 0) s_err = 1;
        st.bk = 0;
        st.b2k = -1;
      }
      if (lane == 0) {
        const float inc = st.b1 - st.b2 + eps;
        st_person_hi(j, st.bk, st.bk, st.b2k, inc);
        if (u < kBidCache) {
          s_bj[u] = j;
          s_bo[u] = st.bk;
          s_b2k[u] = st.b2k;
          s_binc[u] = inc;
        }
        alarm |= emd_band_alarm(pend_old, pend_inc);
        pend_old = atomicMax(reinterpret_cast<u64 *>(&sc.ostate[st.bk]),
                             ((u64)emd_f2ord(inc) << 32) | (u64)((unsigned)j + 1u));
        pend_inc = inc;
        drawn = atomicAdd(&s_next, 1);
      }
      u = __builtin_amdgcn_readlane(drawn, 0);
    }
    }
    alarm |= emd_band_alarm(pend_old, pend_inc);
    if constexpr (W > 1) {
      if (chg_have) {
        const int pmb = (int)(unsigned)chg_pend;
        if (pmb >= 0)  // bits of a non-negative float order like ints
          atomicMax(reinterpret_cast<int *>(&l_lo[(int)(chg_pend >> 32)].w), pmb);
        chg_have = false;
      }
    }
    int *my_alarm = &s_alarm[it & 1];
    if (alarm) *my_alarm = 1;
    if (t == 0) s_cnt[cur ^ 1] = 0;
#ifdef MVP_EMD_PROFILE
    const long long tp1 = __builtin_readcyclecounter();
    if (lane == 0) s_wbusy[wave] = (int)(tp1 - tp0);
#endif
    // ---------------- all bids of the round are placed
    bool any_alarm;
    if (clustered) {
      // The bids themselves are complete (their atomics have returned); the
      // bidders' hint records are only read after the next draining gather.
      if (!emd_cluster_gather<W, false>(slots, wg, ++epoch, my_alarm, my_alarm, s_gout, &s_abort, same_xcd)) {
        aborted = true;
        break;
      }
      any_alarm = false;
#pragma unroll
      for (int w = 0; w < W; ++w) any_alarm |= s_gout[2 * w] != 0u;
    } else {
      __syncthreads();
      any_alarm = *my_alarm != 0;
    }

    // ---------------- GetMax (emd_cuda.cu:181-194): only when two different
    // increments within the 1e-6 band met on one object this round.  Every
    // bidder inside the band of the object's maximal increment raises the
    // key's bidder field; the increment field stays.
    if (__builtin_expect(any_alarm, 0)) {
#ifdef MVP_EMD_PROFILE
      n_alarm += 1;
#endif
      for (int u = t; u < U; u += kEmdThreads) {
        int j, o;
        float bi;
        if (u < kBidCache) {
          j = s_bj[u]; o = s_bo[u]; bi = s_binc[u];
        } else {
          j = L[u];
          const float4 g = ld_person(j, 1);
          o = __float_as_int(g.x); bi = g.w;
        }
        const u64 key = ld_key(o);
        if (emd_in_band(bi, emd_ord2f((unsigned)(key >> 32))))
          atomicMax(reinterpret_cast<u64 *>(&sc.ostate[o]), (key & 0xFFFFFFFF00000000ull) | (u64)((unsigned)j + 1u));
      }
      if (clustered) {
        if (!emd_cluster_gather<W>(slots, wg, ++epoch, my_alarm, my_alarm, s_gout, &s_abort, same_xcd)) {
          aborted = true;
          break;
        }
      } else {
        __syncthreads();
      }
    }
    if (t == 0) {
      s_alarm[(it + 1) & 1] = 0;  // next round's flag; its writers are a barrier away
      s_next = kEmdWaves;         // list positions 0..15 belong to the waves, the rest are drawn
    }
#ifdef MVP_EMD_PROFILE
    const long long tp2 = __builtin_readcyclecounter();
#endif

    // ---------------- Assign (emd_cuda.cu:196-215)
    const int nxt = cur ^ 1;
    u64 *my_chg = sc.chg + (size_t)wg * kChgCap;
    // Few bidders (the long tail): one or two per WAVE instead of all of them in
    // the lanes of wave 0 -- winners, losers and bound refreshes are divergent
    // paths with their own memory round trips, which a single wave would run
    // one after the other.  (Many bidders: consecutive lanes, conflict-free LDS.)
    const bool spread = U <= 4 * kEmdWaves * 4;
    for (int ub = 0; ub < U; ub += kEmdThreads) {
      const int u = ub + (spread ? lane * kEmdWaves + wave : t);
      if (u >= U) continue;
      int j, o, b2k;
      float bi;
      if (__builtin_expect(u < kBidCache, 1)) {
        j = s_bj[u]; o = s_bo[u]; bi = s_binc[u]; b2k = s_b2k[u];
      } else {
        j = L[u];
        const float4 g = ld_person(j, 1);
        o = __float_as_int(g.x); b2k = __float_as_int(g.z); bi = g.w;
      }
#ifdef MVP_EMD_PROFILE
      const long long ta0 = __builtin_readcyclecounter();
#endif
      const int4 os = ld_ostate(o);
      const float4 oo = ld_obj(o);  // independent of `os`: same round trip
#ifdef MVP_EMD_PROFILE
      if (t == 0 && it >= 100) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        prof_a1 += __builtin_readcyclecounter() - ta0;
        prof_an += 1;
      }
#endif
      if (last || (unsigned)os.x == (unsigned)j + 1u) {  // a loser may read after the winner reset the key to 0
        const int prev = os.z;
        if (!last && prev != -1) {
          // the evicted owner bids again next round, in this workgroup's list
          st_i32(&ass[prev], -1);
          const int pos = atomicAdd(&s_cnt[nxt], 1);
          if (__builtin_expect(pos >= kRecCap, 0)) Lnext[pos] = prev;   // entries below kRecCap live in LDS only
          if (__builtin_expect(pos < kRecCap, 1)) {
            const float4 pa = ld_person(prev, 0);
            const float4 pb = ld_person(prev, 1);
            s_rq[nxt][pos] = pa;
            s_ri[nxt][pos] = make_int4(prev, __float_as_int(pb.y), __float_as_int(pb.z), 0);
          }
        }
#ifdef MVP_EMD_PROFILE
        if (t == 0 && it >= 100) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); prof_a2 += __builtin_readcyclecounter() - ta0; }
#endif
        st_ostate(o, j);
        st_i32(&ass[j], o);
        st_f32(&sc.obj[o].w, oo.w + bi);
        // The leaf's price lower bound only needs a refresh when the object
        // that just got dearer was (one of) the cheapest of its leaf; then the
        // members are re-scanned (prices read while other winners raise them
        // are old or new -- either way a valid bound) and the new bound is
        // broadcast to the other workgroups of the cluster.
        const int c = o >> lshift;
        if (oo.w <= l_lo[c].w) {
          float pm = oo.w + bi;
          const int e0 = c << lshift, e1 = e0 + (1 << lshift);
          // (the first 16 members in ONE round trip: all of them up to 16384 points)
          float pv[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) pv[k] = e0 + k != o ? ld_price(e0 + k) : __builtin_inff();
#pragma unroll
          for (int k = 0; k < 16; ++k) pm = __builtin_fminf(pm, pv[k]);
          for (int s = e0 + 16; s < e1; ++s) pm = __builtin_fminf(pm, s == o ? pm : ld_price(s));
          l_lo[c].w = pm;
#ifdef MVP_EMD_PROFILE
          if (it >= 100) atomicAdd(&s_hist2[0], 1ull << 40);  // refresh count in the high bits
#endif
          if (clustered) {
            const int q = atomicAdd(&s_nchg, 1);
            if (q < kChgCap) {
              const u64 e = ((u64)(unsigned)c << 32) | (u64)__float_as_uint(pm);
              if (same_xcd) __hip_atomic_store(my_chg + q, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              else __hip_atomic_store(my_chg + q, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
      } else {
        // lost: stays in the list, record carried over through LDS
        const int pos = atomicAdd(&s_cnt[nxt], 1);
        if (pos >= kRecCap) Lnext[pos] = j;
        if (pos < kRecCap) {
          float4 pa;
          if (u < kRecCap)
            pa = s_rq[cur][u];
          else
            pa = ld_person(j, 0);
          s_rq[nxt][pos] = pa;
          s_ri[nxt][pos] = make_int4(j, o, b2k, 0);
        }
      }
#ifdef MVP_EMD_PROFILE
      if (t == 0 && it >= 100) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); prof_a3 += __builtin_readcyclecounter() - ta0; }
#endif
    }
    // the nodes' price bounds follow their leaves' (LDS only; a bound that lags a round is still a bound)
    if (t >= kEmdThreads - kMaxNodes) emd_node_price(l_lo, n_lo, t - (kEmdThreads - kMaxNodes), nleaf);
#ifdef MVP_EMD_PROFILE
    const long long tp3 = __builtin_readcyclecounter();
    if (t == 0 && it >= 100) prof_a4 += tp3 - tp2;
#endif
    // ---------------- end of round: next list sizes + refreshed price bounds
    if (clustered) {
#ifdef MVP_EMD_PROFILE
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const long long tpd = __builtin_readcyclecounter();
#endif
      if (!emd_cluster_gather<W>(slots, wg, ++epoch, &s_cnt[nxt], &s_nchg, s_gout, &s_abort, same_xcd)) {
        aborted = true;
        break;
      }
#ifdef MVP_EMD_PROFILE
      const long long tpg2 = __builtin_readcyclecounter();
      prof_drain += tpd - tp3;
      prof_gather += tpg2 - tpd;
#endif
      Utot = 0;
      bool overflow = false;
      int cntw[W], chgw[W];  // (the same for every lane: kept in scalar registers, the arithmetic on them is SALU work)
#pragma unroll
      for (int w = 0; w < W; ++w) {
        cntw[w] = __builtin_amdgcn_readfirstlane((int)s_gout[2 * w]);
        chgw[w] = __builtin_amdgcn_readfirstlane((int)s_gout[2 * w + 1]);
        Utot += cntw[w];
        if (w != wg) overflow |= chgw[w] > kChgCap;
      }
      if (lean) {
        // ---- hand the cloud to the lean kernel (emd_lean.hip) once no member runs the
        // four-bidders-per-wave schedule any more: every member leaves its list (<= kRowModeMin
        // entries, all in LDS) at the start of its list area; the rest of the auction state is
        // in the scratch already.  The decision is the same for every member (gathered counts).
        int maxc = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) maxc = max(maxc, cntw[w]);
        if (__builtin_expect(Utot > 0 && Utot <= kLeanCap && maxc <= kRowModeMin && iters - (it + 1) >= lean, 0)) {
          if (t < cntw[wg]) my_ulist[t] = s_ri[nxt][t].x;
          if (t == 0) {
            atomicAdd(reinterpret_cast<unsigned long long *>(&stats[1]), (unsigned long long)n_bids);
            if (s_err) resume->err = 1;   // the second kernel reports it
            resume->cnt[wg] = cntw[wg];
            if (wg == 0) {
              resume->utot = Utot;
              resume->nlists = W;
              resume->first_it = it + 1;
              resume->next_it = it + 1;
              stats[0] = n_rounds;
#ifdef MVP_EMD_CLOUDTIME
              sc.chg[(size_t)kMaxCluster * kChgCap - 64 + 62] = ((u64)W << 48) | ((u64)Utot << 32) | (u64)(unsigned)(wall_clock64() - ct_a0);
#endif
            }
          }
          return;
        }
      }
      if (__builtin_expect(Utot > 0 && Utot <= kSoloMax && it + 1 < iters, 0)) {
        // ---- hand everything to member 0 (lists of <= kSoloMax persons live
        // in LDS only: publish the person ids; their records are in memory)
        if (wg != 0 && t < cntw[wg]) st_i32(my_ulist + t, s_ri[nxt][t].x);
        if (!emd_cluster_gather<W>(slots, wg, ++epoch, &s_nchg, &s_nchg, s_gout, &s_abort, same_xcd)) {
          aborted = true;
          break;
        }
        if (wg != 0) {
          if (t == 0) atomicAdd(reinterpret_cast<unsigned long long *>(&stats[1]), (unsigned long long)n_bids);
          return;
        }
        int idx = t;
#pragma unroll
        for (int w = 1; w < W; ++w) {
          if (idx >= 0 && idx < cntw[w]) {
            const int jj = __hip_atomic_load(sc.ulist + (size_t)w * 2 * n + idx, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
            const float4 pa = ld_person(jj, 0);
            const float4 pb = ld_person(jj, 1);
            const int pos = atomicAdd(&s_cnt[nxt], 1);
            s_rq[nxt][pos] = pa;
            s_ri[nxt][pos] = make_int4(jj, __float_as_int(pb.y), __float_as_int(pb.z), 0);
          }
          idx -= cntw[w];
        }
        clustered = false;
        first = 0;
        share = n;
        if (t == 0) s_nchg = 0;
        __syncthreads();
      } else {
        // ---- rebalance: the round lasts as long as the fullest workgroup's
        // bid passes (16 bidders per pass, 64 in row mode).  When an even
        // split would need fewer passes than the fullest list does, lists
        // above their even share hand the surplus (their last entries) to the
        // lists below it.  Everybody derives the same plan from the gathered
        // counts; the ids travel through the donors' dead current lists.
        int maxc = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) maxc = max(maxc, cntw[w]);
        const int even = (Utot + W - 1) / W;
        const int cap = even > kRowModeMin ? (even + 63) / 64 * 64 : (even + 15) / 16 * 16;
        if (__builtin_expect(maxc > cap && it + 1 < iters, 0)) {
          const int base = Utot / W, rem = Utot % W;
          int exc[W], dfc[W], exoff = 0, dfoff = 0, my_exoff = 0, my_dfoff = 0;
#pragma unroll
          for (int w = 0; w < W; ++w) {
            const int tgt = base + (w < rem ? 1 : 0);
            exc[w] = max(0, cntw[w] - tgt);
            dfc[w] = max(0, tgt - cntw[w]);
            if (w == wg) { my_exoff = exoff; my_dfoff = dfoff; }
            exoff += exc[w];
            dfoff += dfc[w];
          }
          (void)my_exoff;
          const int my_exc = exc[wg], my_dfc = dfc[wg], my_cnt = cntw[wg];
          int *dead = my_ulist + (size_t)cur * n;   // this round's list: no longer read
          for (int i = t; i < my_exc; i += kEmdThreads) {
            const int pos = my_cnt - my_exc + i;
            st_i32(dead + i, pos < kRecCap ? s_ri[nxt][pos].x : Lnext[pos]);
          }
          if (!emd_cluster_gather<W>(slots, wg, ++epoch, &s_nchg, &s_nchg, s_gout, &s_abort, same_xcd)) {
            aborted = true;
            break;
          }
          for (int i = t; i < my_dfc; i += kEmdThreads) {
            int p = my_dfoff + i, jj = -1;   // p-th entry of the pool = donors' surpluses in order
#pragma unroll
            for (int w = 0; w < W; ++w) {
              if (p >= 0 && p < exc[w])
                jj = __hip_atomic_load(sc.ulist + (size_t)w * 2 * n + (size_t)cur * n + p, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
              p -= exc[w];
            }
            const int pos = my_cnt + i;
            if (pos < kRecCap) {
              const float4 pa = ld_person(jj, 0);
              const float4 pb = ld_person(jj, 1);
              s_rq[nxt][pos] = pa;
              s_ri[nxt][pos] = make_int4(jj, __float_as_int(pb.y), __float_as_int(pb.z), 0);
            } else {
              Lnext[pos] = jj;
            }
          }
          if (t == 0) s_cnt[nxt] = my_cnt - my_exc + my_dfc;
          __syncthreads();
#ifdef MVP_EMD_PROFILE
          n_rebal += 1;
#endif
        }
      }
#ifdef MVP_EMD_PROFILE
      const long long tpb = __builtin_readcyclecounter();
      prof_pg1 += tpb - tpg2;
#endif
      if (!clustered) {
        // (just collapsed: nothing to fetch)
      } else if (__builtin_expect(overflow, 0)) {
        // too many refreshes to broadcast (the first, heavy rounds): recompute
        // every bound from the prices themselves (stable between barriers)
        for (int c = t; c < nleaf; c += kEmdThreads) {
          float pm = __builtin_inff();
          for (int s = c << lshift; s < ((c + 1) << lshift); ++s) pm = __builtin_fminf(pm, ld_obj(s).w);
          if (pm >= l_lo[c].w) l_lo[c].w = pm;
        }
      } else {
        // Fetch the other workgroups' refreshed bounds now, fold them in after
        // this workgroup's next bids (a bound that arrives a round late is
        // still a bound): the load latency hides behind the Bid phase.  The
        // producer rewrites its buffer only after the next gather, which this
        // workgroup enters after consuming the value.
        int idx = t, total = 0;
        bool over = false;   // more entries than threads: take them synchronously
#pragma unroll
        for (int w = 0; w < W; ++w) {
          if (w == wg) continue;
          const int cnt = chgw[w];
          if (!chg_have && idx >= 0 && idx < cnt) {
            chg_pend = __hip_atomic_load(sc.chg + (size_t)w * kChgCap + idx, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
            chg_have = true;
          }
          idx -= cnt;   // negative once this thread's entry has been found
          total += cnt;
        }
        over = total > kEmdThreads;
        if (__builtin_expect(over, 0)) {
#pragma unroll
          for (int w = 0; w < W; ++w) {
            if (w == wg) continue;
            const int cnt = chgw[w];
            const u64 *src = sc.chg + (size_t)w * kChgCap;
            for (int i = t; i < cnt; i += kEmdThreads) {
              const u64 e = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              const int pmb = (int)(unsigned)e;
              if (pmb >= 0)  // bits of a non-negative float order like ints
                atomicMax(reinterpret_cast<int *>(&l_lo[(int)(e >> 32)].w), pmb);
            }
          }
          chg_have = false;
          __syncthreads();
        }
      }
      if (t == 0) s_nchg = 0;
    } else {
      __syncthreads();
      Utot = s_cnt[nxt];
      // (a cluster that collapsed to member 0 -- <= kSoloMax persons -- finishes here)
      if (__builtin_expect(W == 1 && lean && Utot > 0 && Utot <= kRowModeMin && iters - (it + 1) >= lean, 0)) {
        if (t < Utot) my_ulist[t] = s_ri[nxt][t].x;
        if (t == 0) {
          if (s_err) resume->err = 1;
          resume->cnt[0] = Utot;
          resume->utot = Utot;
          resume->nlists = 1;
          resume->first_it = it + 1;
          resume->next_it = it + 1;
          stats[0] = n_rounds;
          atomicAdd(reinterpret_cast<unsigned long long *>(&stats[1]), (unsigned long long)n_bids);
        }
        return;
      }
    }
#ifdef MVP_EMD_PROFILE
    const long long tp4 = __builtin_readcyclecounter();
    cyc_bid += tp1 - tp0; cyc_sync1 += tp2 - tp1; cyc_assign += tp3 - tp2; cyc_sync2 += tp4 - tp3;
    if (t == 0 && it >= 100 && U <= kRowModeMin) {
      int mx = 0, sm = 0;
      for (int w = 0; w < kEmdWaves; ++w) { mx = max(mx, s_wbusy[w]); sm += s_wbusy[w]; }
      s_hist[13] += mx; s_hist[14] += sm / kEmdWaves; s_hist[15] += 1; prof_u += U;
    }
#endif
    cur ^= 1;
  }

#ifdef MVP_EMD_PROFILE
  if (clustered && !aborted) {  // cost of the bare all-gather
    const long long tg0 = __builtin_readcyclecounter();
    for (int g = 0; g < 256; ++g)
      if (!emd_cluster_gather<W>(slots, wg, ++epoch, &s_nchg, &s_nchg, s_gout, &s_abort, same_xcd)) break;
    if (t == 0 && cloud < 2 && wg == 0)
      printf("cloud %d: bare cluster all-gather %lld cycles each (W = %d, same_xcd %d)\n", cloud,
             (__builtin_readcyclecounter() - tg0) / 256, W, (int)same_xcd);
  }
#endif
  if (aborted) {
    // A cluster wait ran into its bound (the members were not co-resident for
    // tens of seconds).  Fail loudly: NaN distances, -1 assignments.
    if (t == 0) stats[0] = -2;
    for (int j = t; j < n; j += kEmdThreads) {
      dist[j] = __builtin_nanf("");
      ass[j] = -1;
    }
    return;
  }
  if (t == 0) {
    if (wg == 0) stats[0] = s_err ? -1 : n_rounds;
    if (s_err) stats[0] = -1;
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
        printf("cloud 0 wg 0 searches %s 10k cycles: %llu | mean sub-box %.0f cells, visited %.1f, visit steps %.2f, extra member iterations %.2f, folds %.1f, seed %.0f cycles, visits %.0f cycles\n",
               k ? ">=" : "<", s_slow[k][0], s_slow[k][1] / c, s_slow[k][2] / c, s_slow[k][3] / c, s_slow[k][4] / c, s_slow[k][5] / c, s_slow[k][6] / c, s_slow[k][7] / c);
      }
    if (cloud < 2)
      printf("cloud %d wg %d Assign (thread 0, %lld samples): loads done at %lld cycles, eviction handled at %lld (sum over winning rounds / all), body done at %lld, phase %lld\n", cloud, wg, prof_an, prof_a1 / (prof_an + 1), prof_a2 / (prof_an + 1), prof_a3 / (prof_an + 1), prof_a4 / (prof_an + 1));
    if (cloud < 2)
      printf("cloud %d wg %d tail rounds %llu: bidders/round %.1f, busiest wave %llu cycles/round, mean wave %llu\n", cloud, wg, s_hist[15],
             (double)prof_u / (double)(s_hist[15] + 1), s_hist[13] / (s_hist[15] + 1), s_hist[14] / (s_hist[15] + 1));
    if (cloud < 2)
      printf("cloud %d wg %d wave-mode bids after round 100: %llu, mean cycles %llu, mean sub-box cells %llu, mean cells visited %llu, linear %llu | <2k %llu <4k %llu <8k %llu <16k %llu <32k %llu <64k %llu <128k %llu more %llu\n",
             cloud, wg, s_hist[10], s_hist[12] / (s_hist[10] + 1), s_hist[8] / (s_hist[10] + 1), s_hist[9] / (s_hist[10] + 1), s_hist[11],
             s_hist[0], s_hist[1], s_hist[2], s_hist[3], s_hist[4], s_hist[5], s_hist[6], s_hist[7]);
    if (cloud < 2)
      printf("cloud %d wg %d: rounds %lld bids %lld alarms %lld rebalances %lld | cycles bid %lld sync1 %lld assign %lld sync2 %lld \n",
             cloud, wg, n_rounds, n_bids, n_alarm, n_rebal, cyc_bid, cyc_sync1, cyc_assign, cyc_sync2);
    if (cloud < 2)
      printf("cloud %d wg %d: sync2 = store drain %lld + closing gather %lld + list bookkeeping %lld + bound fetch (rest)\n", cloud, wg, prof_drain, prof_gather, prof_pg1);
#endif
  }
  // ---------------- CalcDist (emd_cuda.cu:217-226); slots -> object indices
  __syncthreads();
  for (int j = first + t; j < first + share; j += kEmdThreads) {
    int s;
    if constexpr (W == 1) s = ass[j];
    else s = __hip_atomic_load(&ass[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float4 o = sc.obj[s];  // coordinates never change
    const float dx = xyz1[j * 3 + 0] - o.x;
    const float dy = xyz1[j * 3 + 1] - o.y;
    const float dz = xyz1[j * 3 + 2] - o.z;
    dist[j] = sqdist3(dx, dy, dz);
    ass[j] = sc.perm[s];
  }
}

// emd NmDistanceGradKernel (emd_cuda.cu:284-300): each (cloud, j) has a single
// writer, so the reference's atomicAdd is a plain accumulate here.
__global__ __launch_bounds__(256) void emd_grad_kernel(
    int n, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
    const float *__restrict__ grad_dist, const int *__restrict__ idx,
    float *__restrict__ grad_xyz) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int cloud = blockIdx.y;
  const size_t a = ((size_t)cloud * n + j) * 3;
  const int j2 = idx[(size_t)cloud * n + j];
  if (j2 < 0 || j2 >= n) return;  // -1: the forward pass was abandoned (NaN distances); no gradient
  const size_t c = ((size_t)cloud * n + j2) * 3;
  const float g = grad_dist[(size_t)cloud * n + j] * 2;
  grad_xyz[a + 0] += g * (xyz1[a + 0] - xyz2[c + 0]);
  grad_xyz[a + 1] += g * (xyz1[a + 1] - xyz2[c + 1]);
  grad_xyz[a + 2] += g * (xyz1[a + 2] - xyz2[c + 2]);
}

// Tuning / A-B knobs of the auction.  Process-wide; compiled-in defaults, mvp_emd_configure() changes them at run time
// (cluster, same_xcd, split, resident_cap).  The release library reads NOTHING from the environment; a build with
// -DMVP_TEST_HOOKS (`make hooks` -> libmvpops_hooks.so, loaded by the tests / tools that need it) also takes, ONCE at first use,
//   MVP_EMD_CLUSTER / MVP_EMD_SAME_XCD / MVP_EMD_SPLIT / MVP_EMD_RESIDENT_CAP   the same four knobs
//   MVP_EMD_PLAN_ROUND / MVP_EMD_PLAN_EVERY / MVP_EMD_PLAN_WIDTHS               the tiered launch's plan (emd_lean.hip)
//   MVP_EMD_TIERS_FAIL                                                          "the tiered kernel does not fit this device"
// split: 0 one kernel runs every round; 1: the rounds after the last four-bidders-per-wave round run in the lean kernel
// (emd_lean.hip); 2: + cluster widths dealt out by load at round 300 (33..64 clouds of >= 4096 points); 3: + clouds of
// <= 4096 points finish LDS-resident on one workgroup, in a launch of their own (emd_resident.hip); 4: ... inside the lean
// launch, on member 0 of the cloud's cluster; 5 (default): + gathered-bid rounds.
// The results do not depend on any of them (bit-identical; tests/test_gpu_ops.py).
struct EmdKnobs {
  int cluster, same_xcd, split;   // split: 0 one kernel, 1 + the lean kernel, 2 + planned cluster widths (emd_lean.hip), 3 + resident tail (own launch), 4 fused into the lean launch, 5 + gathered-bid rounds
  int plan_round, plan_every;     // split == 2: round of the first plan; rounds per planned launch
  unsigned long long plan_widths; // widths of an XCD's 8 cloud slots, heaviest first, 4 bits each
  int res_cap;                    // split == 3: unassigned persons at which a cloud of <= 4096 points moves into LDS (emd_resident.hip)
};
static std::mutex g_knob_mutex;
static EmdKnobs &emd_knobs_locked() {   // (callers hold g_knob_mutex)
  static EmdKnobs k = [] {
    EmdKnobs v{kMaxCluster, 1, 5, 300, 4096, 0ull, 16};   // (mvp_emd_forward_plan builds the same defaults) widths from the loads (MVP_EMD_PLAN_WIDTHS=8,5,4,4,3,3,3,2 fixes them)
#ifdef MVP_TEST_HOOKS
    if (const char *e = getenv("MVP_EMD_CLUSTER")) v.cluster = atoi(e);
    if (const char *e = getenv("MVP_EMD_SAME_XCD")) v.same_xcd = atoi(e) != 0;
    if (const char *e = getenv("MVP_EMD_SPLIT")) v.split = atoi(e) < 0 ? 0 : atoi(e) > 5 ? 5 : atoi(e);
    if (const char *e = getenv("MVP_EMD_RESIDENT_CAP")) v.res_cap = atoi(e) < 1 ? 1 : atoi(e) > kResList ? kResList : atoi(e);
    if (const char *e = getenv("MVP_EMD_PLAN_ROUND")) v.plan_round = atoi(e) < 1 ? 1 : atoi(e);
    if (const char *e = getenv("MVP_EMD_PLAN_EVERY")) v.plan_every = atoi(e) < 64 ? 64 : atoi(e);
    if (const char *e = getenv("MVP_EMD_PLAN_WIDTHS")) {   // e.g. 8,6,4,4,3,3,2,2
      unsigned long long pat = 0ull;
      int s = 0;
      for (const char *q = e; *q && s < 16; ++s) {
        pat |= (unsigned long long)(atoi(q) & 15) << (4 * s);
        while (*q && *q != ',') ++q;
        if (*q == ',') ++q;
      }
      v.plan_widths = pat;
    }
#endif
    return v;
  }();
  return k;
}
// One consistent copy of the knobs (a call of mvp_emd_forward takes it once).
static EmdKnobs emd_knobs() {
  std::lock_guard<std::mutex> g(g_knob_mutex);
  return emd_knobs_locked();
}

// Workgroups per cloud: as many (1, 2, 4, 8) as keep b*W workgroups co-resident,
// one per CU.  MVP_EMD_CLUSTER=1|2|4|8 overrides (still capped by the CU count).
static int emd_cluster_width(int b, int want) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return 1;
  int w = 1;
  while (w * 2 <= want && w * 2 <= kMaxCluster && (long long)b * w * 2 <= cus) w *= 2;
  return w;
}

template <int W>
static hipError_t emd_launch(int b, int n, const float *xyz1, const float *xyz2, float *dist,
                             int *assignment, float eps, int iters, char *scratch, int lean, int fast_ok,
                             hipStream_t stream) {
  int bpad = W == 1 ? b : (b + 7) / 8 * 8;
  if (W == 1) {
    hipLaunchKernelGGL(emd_auction_kernel<1>, dim3(b), dim3(kEmdThreads), 0, stream, b, bpad, n,
                       xyz1, xyz2, dist, assignment, eps, iters, scratch, 0, lean);
    return hipSuccess;
  }
  // cluster members wait for each other: the launch must be checked against
  // the device's residency (cooperative launch does exactly that)
  void *args[] = {&b, &bpad, &n, &xyz1, &xyz2, &dist, &assignment, &eps, &iters, &scratch, &fast_ok, &lean};
  return hipLaunchCooperativeKernel(reinterpret_cast<const void *>(emd_auction_kernel<W>),
                                    dim3(W * bpad), dim3(kEmdThreads), args, 0, stream);
}

}  // namespace mvp

using namespace mvp;

extern "C" long long mvp_emd_scratch_bytes(int b, int n) {
  if (b < 0 || n < 0) return -1;
  return (long long)b * ((long long)emd_scratch_per_cloud(n) + (long long)kEmdTailPerCloud);  // a multiple of 16
}

extern "C" int mvp_emd_configure(int cluster, int same_xcd, int split, int resident_cap) {
  std::lock_guard<std::mutex> g(g_knob_mutex);
  EmdKnobs &k = emd_knobs_locked();
  if (resident_cap == 0 || resident_cap > kResList) return MVP_EBADARG;
  if (resident_cap > 0) k.res_cap = resident_cap;
  if (cluster >= 0) {
    if (cluster != 0 && cluster != 1 && cluster != 2 && cluster != 4 && cluster != 8) return MVP_EBADARG;
    k.cluster = cluster == 0 ? kMaxCluster : cluster;
  }
  if (same_xcd >= 0) k.same_xcd = same_xcd != 0;
  if (split >= 0) k.split = split > 5 ? 5 : split;
  return MVP_OK;
}

static int emd_forward_with(const EmdKnobs &knobs, int b, int n, const float *xyz1,
                            const float *xyz2, float *dist, int *assignment,
                            float eps, int iters, void *scratch,
                            long long scratch_bytes, void *stream) {
  if (b < 0 || n <= 0 || iters < 1) return MVP_EBADSHAPE;
  if (b > 512 || n % 1024 != 0) return MVP_EBADSHAPE;  // emd_cuda.cu:236-249
  // the pruning bounds rely on prices that never fall, i.e. on positive bid increments
  if (!(eps > 0.f)) return MVP_EBADARG;
  if (n > (1 << 20)) return MVP_EBADSHAPE;
  if (b == 0) return MVP_OK;
  if (!xyz1 || !xyz2 || !dist || !assignment || !scratch) return MVP_EBADARG;
  if (scratch_bytes < mvp_emd_scratch_bytes(b, n)) return MVP_EBADARG;
  if ((reinterpret_cast<uintptr_t>(scratch) & 15) != 0) return MVP_EBADARG;
  // The per-cloud areas, barrier granules, hand-over records and statistics take the END of the
  // buffer, so the statistics are its last 16*b bytes however large it is.
  char *sbase = reinterpret_cast<char *>(scratch) + ((scratch_bytes - mvp_emd_scratch_bytes(b, n)) & ~15LL);
  hipStream_t st = as_stream(stream);
  // barrier granules, hand-over records and statistics start from zero on every call
  if (hipMemsetAsync(sbase + (size_t)b * emd_scratch_per_cloud(n), 0, (size_t)b * kEmdTailPerCloud, st) != hipSuccess)
    return check_launch("mvp_emd_forward");
  int w = emd_cluster_width(b, knobs.cluster);
  // Two launches when the auction is long enough to have a tail: the first kernel hands a cloud
  // over when at least `lean` rounds are left (0: never); the second exits at once for clouds
  // that were not handed over.
  const int lean = knobs.split && iters > kLeanMinRounds ? kLeanMinRounds : 0;
  hipError_t err = hipErrorUnknown;
  if (w == 8) err = emd_launch<8>(b, n, xyz1, xyz2, dist, assignment, eps, iters, sbase, lean, knobs.same_xcd, st);
  else if (w == 4) err = emd_launch<4>(b, n, xyz1, xyz2, dist, assignment, eps, iters, sbase, lean, knobs.same_xcd, st);
  else if (w == 2) err = emd_launch<2>(b, n, xyz1, xyz2, dist, assignment, eps, iters, sbase, lean, knobs.same_xcd, st);
  if (err != hipSuccess) {  // w == 1, or the cluster does not fit this device
    (void)hipGetLastError();
    w = 1;
    (void)emd_launch<1>(b, n, xyz1, xyz2, dist, assignment, eps, iters, sbase, lean, 0, st);
  }
  if (lean && emd_lean_launch(b, n, w, xyz1, dist, assignment, eps, iters, sbase, knobs.same_xcd | (knobs.split >= 5 ? 2 : 0),
                              knobs.plan_round, knobs.split >= 2 ? knobs.plan_every : 0, knobs.plan_widths,
                              knobs.split >= 4 ? knobs.res_cap : knobs.split == 3 ? -knobs.res_cap : 0, st) != hipSuccess)
    return check_launch("mvp_emd_forward");
  return check_launch("mvp_emd_forward");
}

extern "C" int mvp_emd_forward(int b, int n, const float *xyz1,
                               const float *xyz2, float *dist, int *assignment,
                               float eps, int iters, void *scratch,
                               long long scratch_bytes, void *stream) {
  return emd_forward_with(emd_knobs(), b, n, xyz1, xyz2, dist, assignment, eps, iters, scratch, scratch_bytes, stream);
}

// The same call with its launch plan on the call instead of in the process (ABI 17): nothing of the library's state is
// read or written -- a NULL plan or a negative field means the compiled-in default (NOT what mvp_emd_configure set).
extern "C" int mvp_emd_forward_plan(int b, int n, const float *xyz1,
                                    const float *xyz2, float *dist, int *assignment,
                                    float eps, int iters, void *scratch,
                                    long long scratch_bytes, const MvpEmdPlan *plan, void *stream) {
  EmdKnobs k{kMaxCluster, 1, 5, 300, 4096, 0ull, 16};
  if (plan) {
    if (plan->cluster >= 0) {
      if (plan->cluster != 0 && plan->cluster != 1 && plan->cluster != 2 && plan->cluster != 4 && plan->cluster != 8) return MVP_EBADARG;
      k.cluster = plan->cluster == 0 ? kMaxCluster : plan->cluster;
    }
    if (plan->same_xcd >= 0) k.same_xcd = plan->same_xcd != 0;
    if (plan->split >= 0) k.split = plan->split > 5 ? 5 : plan->split;
    if (plan->resident_cap == 0 || plan->resident_cap > kResList) return MVP_EBADARG;
    if (plan->resident_cap > 0) k.res_cap = plan->resident_cap;
  }
  return emd_forward_with(k, b, n, xyz1, xyz2, dist, assignment, eps, iters, scratch, scratch_bytes, stream);
}

extern "C" int mvp_emd_backward(int b, int n, const float *xyz1,
                                const float *xyz2, float *gradxyz,
                                const float *graddist, const int *idx,
                                void *stream) {
  if (b < 0 || n < 0) return MVP_EBADSHAPE;
  if (b == 0 || n == 0) return MVP_OK;
  if (!xyz1 || !xyz2 || !gradxyz || !graddist || !idx) return MVP_EBADARG;
  if (b > 65535) return MVP_EBADSHAPE;
  dim3 grid((n + 255) / 256, b);
  hipLaunchKernelGGL(emd_grad_kernel, grid, dim3(256), 0, as_stream(stream), n,
                     xyz1, xyz2, graddist, idx, gradxyz);
  return check_launch("mvp_emd_backward");
}
