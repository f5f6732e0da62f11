// The list-driven tail of the EMD auction: one workgroup per cloud, prices AND
// owners in LDS, the bid search a linear scan of a static per-person list.
// Opt-in (mvp_emd_configure(tail = 2) / MVP_EMD_TAIL=2): bit-exact, its rounds
// are 2-2.4x cheaper than the clustered kernel's once few persons are left, but
// building the lists costs more than that saves (DESIGN.md section 5c,
// profiles/r2b_emd_list_tail.txt).
//
// Same reference semantics as emd.hip / emd_tail.hip (emd_cuda.cu:95-215: Bid,
// GetMax, Assign per round; results bit-identical to the exhaustive oracle),
// for the rounds after the clustered kernel handed the cloud over (at most
// MVP_EMD_SOLO_CAP <= kTailCap = 256 persons unassigned, their number never
// grows).  What a round costs there is a latency chain: the search of the
// slowest bidder (700-1000 dependent instructions of grid enumeration in the
// other two kernels), then the exchange between the workgroups of a cluster.
// Here there is neither:
//
//   * ONE workgroup owns the cloud (no exchange); the prices, the owner of
//     every object and the pool of unassigned persons live in its LDS;
//   * the search is replaced by a scan of a STATIC list.  Between the two
//     kernels `emd_list_build_kernel` writes, for every person of the cloud,
//     the objects in increasing order of  key = fl(sqrtf distance + price at
//     hand-over), grouped in shells of width 1/64 (emd_common.h, kListRec): up
//     to 1024 entries {slot, sqrtf distance}.  Prices only rise, so at any
//     later time an object's value  float(3.0 - sqrtf(s) - price)  is at most
//     3 - key: once the second best of the scanned shells exceeds 3 - (lower
//     edge of the next shell) the scan has provably seen the best and the
//     second best object (and everything within `delta` of the latter, which
//     refills the candidate cache).  The distances are the very floats the
//     search would compute (same expression), so every value is bit-identical.
//     Phase B: one wave per miss; header + 256 entries arrive in one round trip
//     (the next miss's loads already in flight), prices gathered from LDS; per
//     lane only the two best VALUES (min / max) and the slot of the best are
//     kept, two DPP max reductions give the wave's best and second best;
//   * what a list cannot decide goes to phase C, a filtered scan of ALL objects
//     by the whole workgroup with the exact tie logic: lists cut too early for
//     the bid (1024 entries / 31 shells; 0.003-0.5 % of the bids at the
//     headline shape) and ties for the best value (the reference's order on
//     original indices decides them);
//   * exact candidate cache per person as in emd_tail.hip (phase A, four
//     bidders per wave, LDS only), refilled by the scan with the first 16
//     entries worth >= (seed - delta) -- list order is value order at hand-over
//     -- and bounded by max(second best - delta, best value that found no room);
//   * GetMax through the object's owner word; Assign by the bid's own thread,
//     the cache of the person it evicts fetched speculatively before GetMax is
//     resolved; the order list of the pool is rebuilt only when an entry died.
#include "emd_common.h"

namespace mvp {

__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ------------------------------------------------------------------ cell records
// One thread per grid cell: exact bounding box of the members (rounded outwards
// to fp16) and their cheapest price, for the builder's cell test.
__global__ __launch_bounds__(256) void emd_list_cells_kernel(int b, int n, char *scratch, char *lists) {
  const int cloud = blockIdx.y;
  const size_t per_cloud = emd_scratch_per_cloud(n);
  char *tail = scratch + (size_t)b * per_cloud;
  const EmdResume *resume = emd_resume(tail, b, cloud);
  if (resume->next_it == 0) return;
  const int g = resume->g;
  const int ncell = g * g * g;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (ncell > kTailCells || c >= ncell) return;
  const EmdScratch sc = emd_carve(scratch + (size_t)cloud * per_cloud, n);
  uint4 *rec = reinterpret_cast<uint4 *>(lists + (size_t)cloud * emd_lists_per_cloud(n));
  float bx0 = __builtin_inff(), by0 = __builtin_inff(), bz0 = __builtin_inff();
  float bx1 = -__builtin_inff(), by1 = -__builtin_inff(), bz1 = -__builtin_inff();
  float pm = __builtin_inff();
  const int s0 = sc.cstart[c], s1 = sc.cstart[c + 1];
  for (int s = s0; s < s1; ++s) {
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
  r.w = __float_as_uint(s0 < s1 ? pm : 0.f);
  rec[c] = r;
}

// ------------------------------------------------------------------ persons by cell
// One workgroup per cloud: the persons counting-sorted by the grid cell of their point (the
// grid of the objects), so that the builder can share one candidate set among the persons of
// a cell.  pstart[c] .. pstart[c + 1]: positions in pperm (person ids).
__global__ __launch_bounds__(kEmdThreads) void emd_list_persons_kernel(int b, int n, char *scratch, char *lists) {
  const int cloud = blockIdx.x;
  const size_t per_cloud = emd_scratch_per_cloud(n);
  char *tail = scratch + (size_t)b * per_cloud;
  const EmdResume *resume = emd_resume(tail, b, cloud);
  if (resume->next_it == 0) return;
  GridGeom gg;
  gg.g = resume->g;
  gg.lox = resume->lox;
  gg.loy = resume->loy;
  gg.loz = resume->loz;
  gg.invh = resume->invh;
  const int ncell = gg.g * gg.g * gg.g;
  if (ncell > kTailCells) return;
  const EmdScratch sc = emd_carve(scratch + (size_t)cloud * per_cloud, n);
  char *lbase = lists + (size_t)cloud * emd_lists_per_cloud(n);
  unsigned short *pstart = reinterpret_cast<unsigned short *>(lbase + (size_t)(kTailCells + 13) * 16);
  unsigned short *pperm = pstart + (kTailCells + 13);
  __shared__ int cnt[kTailCells + 13], cur[kTailCells + 13];
  const int t = threadIdx.x;
  for (int c = t; c < kTailCells + 13; c += kEmdThreads) cnt[c] = 0;
  __syncthreads();
  for (int j = t; j < n; j += kEmdThreads) {
    const float4 q = sc.person[2 * j];
    atomicAdd(&cnt[emd_cell(gg, q.x, q.y, q.z)], 1);
  }
  __syncthreads();
  if (t < kWave) {  // exclusive prefix: 21 cells per lane
    constexpr int kPer = (kTailCells + 13) / kWave;
    int loc = 0;
    for (int k = 0; k < kPer; ++k) loc += cnt[t * kPer + k];
    int inc = loc;
    for (int d = 1; d < kWave; d <<= 1) {
      const int o = __shfl_up(inc, d, kWave);
      if (t >= d) inc += o;
    }
    int run = inc - loc;
    for (int k = 0; k < kPer; ++k) {
      const int v = cnt[t * kPer + k];
      cur[t * kPer + k] = run;
      run += v;
    }
  }
  __syncthreads();
  for (int c = t; c <= ncell; c += kEmdThreads) pstart[c] = (unsigned short)(c < ncell ? cur[c] : n);
  __syncthreads();
  for (int j = t; j < n; j += kEmdThreads) {
    const float4 q = sc.person[2 * j];
    const int pos = atomicAdd(&cur[emd_cell(gg, q.x, q.y, q.z)], 1);
    pperm[pos] = (unsigned short)j;
  }
}

// ------------------------------------------------------------------ list builder
// One workgroup per PERSON cell.  The candidate objects of all its persons are chosen once,
// from cell-level bounds: for an object cell, (distance between the two bounding boxes +
// cheapest price) is a lower bound of the key of every (person of this cell, member) pair --
// float subtraction, multiply-add, sqrt and add are monotone, so it bounds the COMPUTED keys.
// Cells are histogrammed by the shell of that bound, weighted with their population; the
// cells of the largest shell S whose cumulated population fits kListObj are copied into LDS.
// Then one wave per person: pass 1 histograms the exact shells of the staged objects, the
// leading shells that fit kListCap entries (and are <= S, i.e. complete) are kept; pass 2
// places {slot, sqrtf distance} at its shell's running position; the list leaves with
// coalesced stores.
constexpr int kBuildWaves = 8;
constexpr int kListObj = 5632;  // staged objects per person cell

__global__ __launch_bounds__(kBuildWaves *kWave) void emd_list_build_kernel(int b, int n, char *scratch, char *lists) {
  const int cloud = blockIdx.y;
  const size_t per_cloud = emd_scratch_per_cloud(n);
  char *tail = scratch + (size_t)b * per_cloud;
  const EmdResume *resume = emd_resume(tail, b, cloud);
  if (resume->next_it == 0) return;  // finished in the clustered kernel (uniform over the block)
  const int g = resume->g;
  const int ncell = g * g * g;
  const int pc = blockIdx.x;
  if (ncell > kTailCells || pc >= ncell) return;
  const EmdScratch sc = emd_carve(scratch + (size_t)cloud * per_cloud, n);
  char *lbase = lists + (size_t)cloud * emd_lists_per_cloud(n);
  const uint4 *rec = reinterpret_cast<const uint4 *>(lbase);
  const unsigned short *pstart = reinterpret_cast<const unsigned short *>(lbase + (size_t)(kTailCells + 13) * 16);
  const unsigned short *pperm = pstart + (kTailCells + 13);
  const int p0 = pstart[pc], p1 = pstart[pc + 1];
  if (p0 >= p1) return;

  __shared__ float4 s_obj[kListObj];
  __shared__ unsigned short s_oslot[kListObj];
  __shared__ unsigned short out_slot[kBuildWaves][kListCap];
  __shared__ float out_d[kBuildWaves][kListCap];
  __shared__ unsigned short wl[kTailCells + 13], wlo[kTailCells + 13];
  __shared__ int hist[kBuildWaves][2][32];
  __shared__ int hist0[32];
  __shared__ unsigned s_box[6];
  __shared__ int s_nl, s_no, s_senum;

  const int t = threadIdx.x, lane = t & (kWave - 1), wave = t >> 6;
  const int row = t >> 4, l16 = t & 15;  // 32 rows of 16 lanes
  if (t < 32) hist0[t] = 0;
  if (t < 6) s_box[t] = t < 3 ? 0xFFFFFFFFu : 0u;
  if (t == 0) {
    s_nl = 0;
    s_no = 0;
  }
  __syncthreads();
  // bounding box of this cell's persons (exact)
  for (int p = p0 + t; p < p1; p += kBuildWaves * kWave) {
    const float4 q = sc.person[2 * (int)pperm[p]];
    atomicMin(&s_box[0], emd_f2ord(q.x));
    atomicMin(&s_box[1], emd_f2ord(q.y));
    atomicMin(&s_box[2], emd_f2ord(q.z));
    atomicMax(&s_box[3], emd_f2ord(q.x));
    atomicMax(&s_box[4], emd_f2ord(q.y));
    atomicMax(&s_box[5], emd_f2ord(q.z));
  }
  __syncthreads();
  const float px0 = emd_ord2f(s_box[0]), py0 = emd_ord2f(s_box[1]), pz0 = emd_ord2f(s_box[2]);
  const float px1 = emd_ord2f(s_box[3]), py1 = emd_ord2f(s_box[4]), pz1 = emd_ord2f(s_box[5]);
  // object cells: shell of the bound, population
  int csh[3], ccnt[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int c = t + k * kBuildWaves * kWave;
    csh[k] = 99;
    ccnt[k] = 0;
    if (c < ncell) {
      const int s0 = sc.cstart[c], s1 = sc.cstart[c + 1];
      ccnt[k] = s1 - s0;
      if (s1 > s0) {
        const uint4 cr = rec[c];
        const float dx = __builtin_fmaxf(
            __builtin_fmaxf(half_bits_to_float(cr.x & 0xFFFFu) - px1, px0 - half_bits_to_float(cr.x >> 16)), 0.f);
        const float dy = __builtin_fmaxf(
            __builtin_fmaxf(half_bits_to_float(cr.y & 0xFFFFu) - py1, py0 - half_bits_to_float(cr.y >> 16)), 0.f);
        const float dz = __builtin_fmaxf(
            __builtin_fmaxf(half_bits_to_float(cr.z & 0xFFFFu) - pz1, pz0 - half_bits_to_float(cr.z >> 16)), 0.f);
        const float ks = (__builtin_sqrtf(sqdist3(dx, dy, dz)) + __uint_as_float(cr.w)) * kListScale;
        if (ks < (float)kListShells) {
          csh[k] = (int)ks;
          atomicAdd(&hist0[csh[k]], ccnt[k]);
        }
      }
    }
  }
  __syncthreads();
  if (t < kWave) {
    int cum = 0;
    if (t < kListShells)
      for (int k = 0; k <= t; ++k) cum += hist0[k];
    const unsigned long long okm = __ballot(t < kListShells && cum <= kListObj);
    if (t == 0) s_senum = __builtin_popcountll(okm) - 1;  // (cum is monotone in the lane: a prefix mask)
  }
  __syncthreads();
  const int s_enum = s_senum;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (csh[k] <= s_enum) {
      const int i = atomicAdd(&s_nl, 1);
      wl[i] = (unsigned short)(t + k * kBuildWaves * kWave);
      wlo[i] = (unsigned short)atomicAdd(&s_no, ccnt[k]);
    }
  }
  __syncthreads();
  const int nlist = s_nl, nobj = s_no;
  // their members into LDS: a 16-lane row per cell
  for (int k = row; k < nlist; k += (kBuildWaves * kWave) / 16) {
    const int c = wl[k];
    const int s0 = sc.cstart[c], s1 = sc.cstart[c + 1];
    const int off = wlo[k];
    for (int m = l16; m < s1 - s0; m += 16) {
      s_obj[off + m] = sc.obj[s0 + m];
      s_oslot[off + m] = (unsigned short)(s0 + m);
    }
  }
  __syncthreads();
  // one wave per person
  for (int p = p0 + wave; p < p1; p += kBuildWaves) {
    const int j = pperm[p];
    const float4 q4 = sc.person[2 * j];
    const float qx = q4.x, qy = q4.y, qz = q4.z;
    if (lane < 32) {
      hist[wave][0][lane] = 0;
      hist[wave][1][lane] = 0;
    }
    wave_lds_fence();
    auto shell_of = [&](const float4 &o, float &d) -> int {
      d = __builtin_sqrtf(sqdist3(o.x - qx, o.y - qy, o.z - qz));
      const float ks = (d + o.w) * kListScale;
      return ks < (float)kListShells ? (int)ks : 99;
    };
    // pass 1: exact shell sizes
    for (int i = lane; i < nobj; i += 2 * kWave) {
      const float4 oa = s_obj[i];
      const bool hb = i + kWave < nobj;
      const float4 ob = s_obj[hb ? i + kWave : i];
      float da, db;
      const int sa = shell_of(oa, da), sb = shell_of(ob, db);
      if (sa <= s_enum) atomicAdd(&hist[wave][0][sa], 1);
      if (hb && sb <= s_enum) atomicAdd(&hist[wave][0][sb], 1);
    }
    wave_lds_fence();
    // complete shells that fit the list
    int cum1 = 0, mine = 0;
    if (lane < kListShells) {
      for (int k = 0; k <= lane; ++k) cum1 += hist[wave][0][k];
      mine = hist[wave][0][lane];
    }
    const unsigned long long fit = __ballot(lane < kListShells && lane <= s_enum && cum1 <= kListCap);
    const int ns = __builtin_popcountll(fit);
    const int total = ns > 0 ? __builtin_amdgcn_readlane(cum1, ns - 1) : 0;
    if (lane < 32) hist[wave][1][lane] = cum1 - mine;  // running write position of the shell
    wave_lds_fence();
    // pass 2: the entries, grouped by shell
    for (int i = lane; i < nobj; i += 2 * kWave) {
      const float4 oa = s_obj[i];
      const bool hb = i + kWave < nobj;
      const float4 ob = s_obj[hb ? i + kWave : i];
      float da, db;
      const int sa = shell_of(oa, da), sb = shell_of(ob, db);
      if (sa < ns) {
        const int pos = atomicAdd(&hist[wave][1][sa], 1);
        out_slot[wave][pos] = s_oslot[i];
        out_d[wave][pos] = da;
      }
      if (hb && sb < ns) {
        const int pos = atomicAdd(&hist[wave][1][sb], 1);
        out_slot[wave][pos] = s_oslot[i + kWave];
        out_d[wave][pos] = db;
      }
    }
    wave_lds_fence();
    char *lrec = lbase + kListCellArea + (size_t)j * kListRec;
    unsigned short *hdr = reinterpret_cast<unsigned short *>(lrec);
    unsigned short *oslot = reinterpret_cast<unsigned short *>(lrec + 64);
    float *od = reinterpret_cast<float *>(lrec + 64 + (size_t)kListCap * 2);
    for (int i = lane; i < total; i += kWave) {
      oslot[i] = out_slot[wave][i];
      od[i] = out_d[wave][i];
    }
    if (lane < 32) hdr[lane] = (unsigned short)(lane == 31 ? ns : (lane < ns ? cum1 : total));
#ifdef MVP_EMD_PROFILE
    if (cloud == 0 && lane == 0 && (j % 1024) == 0)
      printf("build person %d: cells staged %d (objects %d) s_enum %d -> shells %d entries %d\n", j, nlist, nobj, s_enum, ns, total);
#endif
    wave_lds_fence();
  }
}

// ------------------------------------------------------------------ the rounds
__global__ __launch_bounds__(kEmdThreads) void emd_solo_kernel(int b, int n, const float *__restrict__ xyz1,
                                                               float *__restrict__ dist, int *assignment, float eps,
                                                               int iters, char *scratch, const char *__restrict__ lists,
                                                               float delta) {
  constexpr int POOL = kTailCap;
  const int cloud = (int)blockIdx.x;
  if (cloud >= b) return;
  const size_t per_cloud = emd_scratch_per_cloud(n);
  char *cbase = scratch + (size_t)cloud * per_cloud;
  char *tail = scratch + (size_t)b * per_cloud;
  EmdResume *resume = emd_resume(tail, b, cloud);
  long long *stats = emd_stats(tail, b, cloud);
  const int it0 = resume->next_it;
  if (it0 == 0) return;  // the cloud was finished by the first kernel

  const int t = threadIdx.x;
  const int lane = t & (kWave - 1);
  const int wave = t >> 6;
  const int row = lane >> 4, l16 = lane & 15, rsh = row * 16;
  xyz1 += (size_t)cloud * n * 3;
  dist += (size_t)cloud * n;
  int *ass = assignment + (size_t)cloud * n;
  const EmdScratch sc = emd_carve(cbase, n);
  const char *lbase = lists + (size_t)cloud * emd_lists_per_cloud(n) + kListCellArea;

  // Mutable global state (assignment, person records, caches) is read with L1-bypassing loads
  // and written through; the lists are static: plain loads.
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(cbase, 0, (int)per_cloud, 0x00020000);
  const unsigned off_cache = (unsigned)(reinterpret_cast<char *>(sc.cache) - cbase);
  auto ldg16 = [&](unsigned byte_off) -> v4u { return __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16); };
  auto stg16 = [&](v4u v, unsigned byte_off) { __builtin_amdgcn_raw_buffer_store_b128(v, rs, byte_off, 0, 16); };
  auto st_i32 = [&](int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto ld_i32 = [&](int *p) -> int { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };

  __shared__ float s_price[kTailMaxN];
  __shared__ unsigned short s_owner[kTailMaxN];             // 0xFFFF: free
  // the pool of unassigned persons
  __shared__ int4 e_i[POOL];                                // {person, hint slot 1, hint slot 2, cached candidates}
  __shared__ __attribute__((aligned(16))) unsigned short e_cs[POOL][kTailK];  // cached slots
  __shared__ __attribute__((aligned(16))) float e_cd[POOL][kTailK];           // their sqrtf distances
  __shared__ float e_tau[POOL];                             // >= value of every object NOT cached
  __shared__ unsigned char e_live[POOL];
  __shared__ unsigned short s_order[POOL];                  // live entries, ascending
  __shared__ unsigned short s_miss[POOL];                   // list positions whose cache failed
  __shared__ float s_seed[POOL];                            // ... and the seed their cache gives
  __shared__ unsigned short s_be[POOL];                     // the bidder's entry
  __shared__ int s_bo[kTailCap], s_b2k[kTailCap], s_bj[kTailCap];  // this round's bids
  __shared__ float s_binc[kTailCap];
  __shared__ int s_gcnt[kTailCap];                          // bids on the object a bid represents
  __shared__ unsigned s_gmax[kTailCap];                     // ... and their maximal increment (ordered bits)
  __shared__ unsigned short s_fall[POOL];                   // bids the lists could not prove
  __shared__ float2 s_wv[kEmdWaves];                        // phase C: the waves' top two
  __shared__ int2 s_wk[kEmdWaves];
  __shared__ int s_nmiss, s_nfall, s_fcnt, s_err, s_U, s_nfree, s_died;
  __shared__ unsigned s_fdmax;
#ifdef MVP_EMD_PROFILE
  // [0] hits [1] misses [2] cycles in misses [3] entries scanned [4] fallbacks [5] cycles in fallbacks [6] unprovable at first
  __shared__ unsigned long long s_prof[16];
  if (threadIdx.x < 16) s_prof[threadIdx.x] = 0ull;
  long long cyc_a = 0, cyc_b = 0, cyc_resolve = 0, cyc_assign = 0, cyc_compact = 0,
            cyc_setup = __builtin_readcyclecounter();
#endif

  const int U0 = resume->utot;
  if (U0 > kTailCap || U0 <= 0) {  // cannot happen
    if (t == 0) stats[0] = -1;
    return;
  }
  // ------------------------------------------------------------ load the state
  for (int s = t; s < n; s += kEmdThreads) {
    s_price[s] = sc.obj[s].w;
    const int ow = sc.ostate[s].z;
    s_owner[s] = (unsigned short)(ow < 0 ? 0xFFFF : ow);
  }
  if (t < POOL) {
    const bool live = t < U0;
    e_live[t] = live ? 1 : 0;
    s_order[t] = (unsigned short)t;
    if (live) {
      const int j = resume->list[t];
      e_i[t] = make_int4(j, -1, -1, 0);  // caches start empty
    }
  }
  if (t == 0) {
    s_nmiss = 0;
    s_nfall = 0;
    s_died = 0;
    s_err = resume->pad;
    s_U = U0;
    s_nfree = 0;
  }
  __syncthreads();
#ifdef MVP_EMD_PROFILE
  cyc_setup = __builtin_readcyclecounter() - cyc_setup;
#endif

  const int block_cnt = n / 1024;
  long long n_rounds = 0, n_bids = 0;
  int U = U0;
  for (int it = it0; it < iters; ++it) {
    if (U == 0) break;
    n_rounds += 1;
    n_bids += U;
    const bool last = it == iters - 1;
    // thread_per_unass of the reference (emd_cuda.cu:107-109): fixes the tie order only
    const int upb = (U + block_cnt - 1) / block_cnt;
    const int tpu = 1024 / upb;
#ifdef MVP_EMD_PROFILE
    const long long tp0 = __builtin_readcyclecounter();
#endif
    // ---------------- Bid, phase A: the cached candidates at today's prices, four bidders per
    // wave (one per 16-lane row).  Exact when the best is > tau and the second best >= tau
    // (every object outside the cache is worth <= tau); a tie for the best goes to the scan.
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
    lds_barrier();
#ifdef MVP_EMD_PROFILE
    const long long tpa = __builtin_readcyclecounter();
#endif

    // ---------------- Bid, phase B (emd_cuda.cu:95-179): the misses, one wave each (wave w takes
    // misses w, w + 16, ...; the loads of its next miss travel while the current one is scanned).
    // Per lane only the two best VALUES and the slot of the best are tracked (min / max, no
    // branches); a tie for the best -- decided by the reference's order on original indices --
    // is detected at the end and sent to phase C.
    const int nmiss = s_nmiss;
    struct Fetched {
      int u, e, j;
      unsigned hw;      // lanes 0..15: hdr[2 lane] | hdr[2 lane + 1] << 16
      unsigned sa, sb;  // slots 2 lane, 2 lane + 1 (| << 16) of the first / second 128 entries
      float2 da, db;    // their distances
    };
    auto issue = [&](int m) -> Fetched {
      Fetched f;
      f.u = s_miss[m];
      f.e = s_be[f.u];
      f.j = e_i[f.e].x;
      const char *lrec = lbase + (size_t)f.j * kListRec;
      f.hw = reinterpret_cast<const unsigned *>(lrec)[lane & 15];
      const unsigned *ls = reinterpret_cast<const unsigned *>(lrec + 64);
      const float2 *ld = reinterpret_cast<const float2 *>(lrec + 64 + (size_t)kListCap * 2);
      f.sa = ls[lane];
      f.sb = ls[kWave + lane];
      f.da = ld[lane];
      f.db = ld[kWave + lane];
      return f;
    };
    int mcur = wave;
    Fetched nx = {};
    if (mcur < nmiss) nx = issue(mcur);
    while (mcur < nmiss) {
      const Fetched cu = nx;
      const int mnext = mcur + kEmdWaves;
      if (mnext < nmiss) nx = issue(mnext);
#ifdef MVP_EMD_PROFILE
      const long long tb0 = __builtin_readcyclecounter();
#endif
      const int u = cu.u, e = cu.e, j = cu.j;
      auto hdr_at = [&](int sidx) -> int {  // hdr[sidx], sidx wave-uniform in 0..31
        const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)cu.hw, sidx >> 1);
        return (int)((sidx & 1) ? (w >> 16) : (w & 0xFFFFu));
      };
      const int ns = min(hdr_at(31), kListShells);
      const int total = ns > 0 ? hdr_at(ns - 1) : 0;
      const char *lrec = lbase + (size_t)j * kListRec;
      const unsigned *ls = reinterpret_cast<const unsigned *>(lrec + 64);
      const float2 *ld = reinterpret_cast<const float2 *>(lrec + 64 + (size_t)kListCap * 2);

      float a1 = -1e9f, a2 = -1e9f;  // the two best values THIS LANE saw (with multiplicity)
      int k1 = -1;                   // slot of a1 (the first to reach it)
      int nst = 0;                   // candidates offered to the cache (wave-uniform)
      float dmax = -1e9f;            // best value that found no room in it
      float thr = -1e9f;
      // two consecutive entries per lane: values, running top two, slot of the best
      auto eval2 = [&](unsigned sw, float2 dd, int idx0, int lim, float &v0, float &v1, int &s0, int &s1) {
        const bool o0 = idx0 < lim, o1 = idx0 + 1 < lim;
        s0 = o0 ? (int)(sw & 0xFFFFu) : 0;
        s1 = o1 ? (int)(sw >> 16) : 0;
        v0 = emd_value_d(dd.x, s_price[s0]);
        v1 = emd_value_d(dd.y, s_price[s1]);
        v0 = o0 ? v0 : -1e9f;
        v1 = o1 ? v1 : -1e9f;
        k1 = v0 > a1 ? s0 : k1;
        top2_insert(a1, a2, v0);
        k1 = v1 > a1 ? s1 : k1;
        top2_insert(a1, a2, v1);
      };
      // The candidate cache is refilled as the scan goes: the first kTailK entries worth
      // >= thr (a list is ordered by the value at hand-over, so these are the likely best) go
      // straight into the pool entry; `dmax` = the best value that found no room.
      auto stage2 = [&](float v0, float v1, int s0, int s1, float2 dd) {
        if (__any(v0 >= thr || v1 >= thr)) {
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const float v = k ? v1 : v0;
            const bool ps = v >= thr && v > -1e8f;
            const unsigned long long mk = __ballot(ps);
            if (ps) {
              const int pos = nst + __builtin_popcountll(mk & ((1ull << lane) - 1ull));
              if (pos < kTailK) {
                e_cs[e][pos] = (unsigned short)(k ? s1 : s0);
                e_cd[e][pos] = k ? dd.y : dd.x;
              } else {
                dmax = __builtin_fmaxf(dmax, v);
              }
            }
            nst += __builtin_popcountll(mk);
          }
        }
      };
      float va0, va1, vb0, vb1;
      int sa0, sa1, sb0, sb1;
      eval2(cu.sa, cu.da, 2 * lane, total, va0, va1, sa0, sa1);
      // the seed: second best of what is known so far (real, distinct objects)
      const float seed = __builtin_fmaxf(s_seed[u], wave_second_largest(a1, a2));
      // shells 0 .. sneed hold everything worth >= seed - delta (see the header): what an
      // unscanned shell s > sneed holds is worth <= 3 - s/64 (+ rounding) < seed - delta
      int sneed = kListShells;
      if (seed > -1e8f) {
        const float rsn = ((3.0f - (seed - delta)) + kMargin) * kListScale;
        sneed = rsn < (float)kListShells ? (int)rsn : kListShells;
      }
      sneed = __builtin_amdgcn_readfirstlane(sneed);
      thr = seed - delta;  // (-1e9: the first kTailK entries are cached)
      const int want = sneed < ns ? hdr_at(sneed & 31) : total;
      stage2(va0, va1, sa0, sa1, cu.da);
      if (want > 2 * kWave) {
        eval2(cu.sb, cu.db, 2 * kWave + 2 * lane, want, vb0, vb1, sb0, sb1);
        stage2(vb0, vb1, sb0, sb1, cu.db);
      }
      for (int base = 4 * kWave; base < want; base += 4 * kWave) {
        const unsigned wa = ls[base / 2 + lane], wb = ls[base / 2 + kWave + lane];
        const float2 fa = ld[base / 2 + lane], fb = ld[base / 2 + kWave + lane];
        eval2(wa, fa, base + 2 * lane, want, va0, va1, sa0, sa1);
        stage2(va0, va1, sa0, sa1, fa);
        if (want > base + 2 * kWave) {
          eval2(wb, fb, base + 2 * kWave + 2 * lane, want, vb0, vb1, sb0, sb1);
          stage2(vb0, vb1, sb0, sb1, fb);
        }
      }
      // the wave's best and second best; a tie for the best is not decided here
      float m1 = a1;
      m1 = __builtin_fmaxf(m1, dpp_f32<0xB1, 0xF>(-1e9f, m1));
      m1 = __builtin_fmaxf(m1, dpp_f32<0x4E, 0xF>(-1e9f, m1));
      m1 = __builtin_fmaxf(m1, dpp_f32<0x141, 0xF>(-1e9f, m1));
      m1 = __builtin_fmaxf(m1, dpp_f32<0x140, 0xF>(-1e9f, m1));
      m1 = __builtin_fmaxf(m1, dpp_f32<0x142, 0xA>(-1e9f, m1));
      m1 = __builtin_fmaxf(m1, dpp_f32<0x143, 0xC>(-1e9f, m1));
      const float b1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m1), 63));
      const unsigned long long w1m = __ballot(a1 == b1);
      const int w1 = __builtin_ctzll(w1m | (1ull << 63));
      const bool tie = (w1m & (w1m - 1)) != 0ull || __any(a1 == b1 && a2 == b1);
      const int bk = __builtin_amdgcn_readlane(k1, w1);
      float m2 = lane == w1 ? a2 : a1;
      m2 = __builtin_fmaxf(m2, dpp_f32<0xB1, 0xF>(-1e9f, m2));
      m2 = __builtin_fmaxf(m2, dpp_f32<0x4E, 0xF>(-1e9f, m2));
      m2 = __builtin_fmaxf(m2, dpp_f32<0x141, 0xF>(-1e9f, m2));
      m2 = __builtin_fmaxf(m2, dpp_f32<0x140, 0xF>(-1e9f, m2));
      m2 = __builtin_fmaxf(m2, dpp_f32<0x142, 0xA>(-1e9f, m2));
      m2 = __builtin_fmaxf(m2, dpp_f32<0x143, 0xC>(-1e9f, m2));
      const float b2 = tie ? b1 : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m2), 63));
      float dm = dmax;
      if (nst > kTailK) {  // (wave-uniform) something found no room in the cache
        dm = __builtin_fmaxf(dm, dpp_f32<0xB1, 0xF>(-1e9f, dm));
        dm = __builtin_fmaxf(dm, dpp_f32<0x4E, 0xF>(-1e9f, dm));
        dm = __builtin_fmaxf(dm, dpp_f32<0x141, 0xF>(-1e9f, dm));
        dm = __builtin_fmaxf(dm, dpp_f32<0x140, 0xF>(-1e9f, dm));
        dm = __builtin_fmaxf(dm, dpp_f32<0x142, 0xA>(-1e9f, dm));
        dm = __builtin_fmaxf(dm, dpp_f32<0x143, 0xC>(-1e9f, dm));
      }
      bool proven = sneed < ns;  // (b2 >= seed: the shells the final b2 needs were scanned)
      if (!proven && ns > 0 && b2 > -1e8f) {
        // every stored shell was scanned: does the result prove itself?
        const float rsn = ((3.0f - (b2 - delta)) + kMargin) * kListScale;
        proven = rsn < (float)ns;  // (int)rsn <= ns - 1
      }
      if (lane == kWave - 1) {
        if (proven && !tie && bk >= 0 && b2 > -1e8f) {
          // everything worth >= b2 - delta was offered to the cache (thr <= b2 - delta);
          // what is not in it -- dropped, scanned or not -- is worth <= tau
          e_tau[e] = __builtin_fmaxf(b2 - delta, dm);
          e_i[e].w = min(nst, kTailK);
          s_bo[u] = bk;
          s_binc[u] = b1 - b2 + eps;
          s_bj[u] = j;
          s_b2k[u] = -1;
        } else {
          // the list was cut too early for this bid, or the best is a tie: phase C
          s_fall[atomicAdd(&s_nfall, 1)] = (unsigned short)u;
          s_seed[u] = b2;  // (a lower bound of the final second best, or -1e9)
        }
      }
#ifdef MVP_EMD_PROFILE
      if (lane == 0) {
        atomicAdd(&s_prof[1], 1ull);
        atomicAdd(&s_prof[2], (unsigned long long)(__builtin_readcyclecounter() - tb0));
        atomicAdd(&s_prof[3], (unsigned long long)want);
      }
#endif
      mcur = mnext;
    }
#ifdef MVP_EMD_PROFILE
    const long long tpg = __builtin_readcyclecounter();
    if (t == 0) s_prof[0] += (unsigned long long)(U - nmiss);
#endif
    lds_barrier();
    // ---------------- Bid, phase C: the bids no list could prove, one at a time, by the whole
    // workgroup -- filtered scan of ALL objects (emd_common.h, kMargin, with B2 := fl(b2 - delta))
    const int nfall = s_nfall;
#ifdef MVP_EMD_PROFILE
    const long long tpc0 = __builtin_readcyclecounter();
#endif
    for (int f = 0; f < nfall; ++f) {
      const int u = s_fall[f];
      const int e = s_be[u];
      const float4 ra = sc.person[2 * e_i[e].x];  // (the point never changes)
      const float qx = ra.x, qy = ra.y, qz = ra.z;
      const float b2lb = s_seed[u];
      const float tm = b2lb > -1e8f ? (3.0f - (b2lb - delta)) + kMargin : 1e9f;
      const float thr2 = b2lb > -1e8f ? b2lb - delta : -1e9f;
      if (t == 0) {
        s_fcnt = 0;
        s_fdmax = 0u;
      }
      lds_barrier();
      float lb1 = -1e9f, lb2 = -1e9f;
      int lbk = -1, lb2k = -1;
      for (int base = 0; base < n; base += 4 * kEmdThreads) {
        float4 o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int sl = base + r * kEmdThreads + t;
          o[r] = sl < n ? sc.obj[sl] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int sl = base + r * kEmdThreads + t;
          if (sl < n) {
            const float p = s_price[sl];
            const float sd = sqdist3(o[r].x - qx, o[r].y - qy, o[r].z - qz);
            const float tq = tm - p;
            if (tq >= 0.f && sd <= tq * tq) {
              const float d = __builtin_sqrtf(sd);
              const float v = emd_value_d(d, p);
              if (v >= thr2) {
                const int pos = atomicAdd(&s_fcnt, 1);
                if (pos < kTailK) {
                  e_cs[e][pos] = (unsigned short)sl;
                  e_cd[e][pos] = d;
                } else {
                  atomicMax(&s_fdmax, emd_f2ord(v));
                }
              }
              if (v > lb1) {
                lb2 = lb1; lb2k = lbk; lb1 = v; lbk = sl;
              } else if (v == lb1) {
                lb2 = v;
                if (emd_precedes(sc.perm[sl], sc.perm[lbk], n, tpu)) {
                  lb2k = lbk; lbk = sl;
                } else {
                  lb2k = sl;
                }
              } else if (v > lb2) {
                lb2 = v; lb2k = sl;
              }
            }
          }
        }
      }
      top2k_merge_step<0xB1, 0xF>(lb1, lbk, lb2, lb2k, n, tpu, sc.perm);
      top2k_merge_step<0x4E, 0xF>(lb1, lbk, lb2, lb2k, n, tpu, sc.perm);
      top2k_merge_step<0x141, 0xF>(lb1, lbk, lb2, lb2k, n, tpu, sc.perm);
      top2k_merge_step<0x140, 0xF>(lb1, lbk, lb2, lb2k, n, tpu, sc.perm);
      top2k_merge_step<0x142, 0xA>(lb1, lbk, lb2, lb2k, n, tpu, sc.perm);
      top2k_merge_step<0x143, 0xC>(lb1, lbk, lb2, lb2k, n, tpu, sc.perm);
      if (lane == kWave - 1) {
        s_wv[wave] = make_float2(lb1, lb2);
        s_wk[wave] = make_int2(lbk, lb2k);
      }
      lds_barrier();
      if (wave == 0) {
        float x1 = -1e9f, x2 = -1e9f;
        int k1 = -1, k2 = -1;
        if (lane < kEmdWaves) {
          x1 = s_wv[lane].x; x2 = s_wv[lane].y;
          k1 = s_wk[lane].x; k2 = s_wk[lane].y;
        }
        top2k_merge_step<0xB1, 0xF>(x1, k1, x2, k2, n, tpu, sc.perm);
        top2k_merge_step<0x4E, 0xF>(x1, k1, x2, k2, n, tpu, sc.perm);
        top2k_merge_step<0x141, 0xF>(x1, k1, x2, k2, n, tpu, sc.perm);
        top2k_merge_step<0x140, 0xF>(x1, k1, x2, k2, n, tpu, sc.perm);
        if (lane == 0) {
          if (k1 < 0) {  // cannot happen (>= 2 objects always survive); never index with -1
            s_err = 1;
            k1 = 0;
            k2 = -1;
          }
          const float dmv = s_fdmax ? emd_ord2f(s_fdmax) : -1e9f;
          e_tau[e] = __builtin_fmaxf(x2 - delta, dmv);
          e_i[e].w = min(s_fcnt, kTailK);
          s_bo[u] = k1;
          s_binc[u] = x1 - x2 + eps;
          s_bj[u] = e_i[e].x;
          s_b2k[u] = k2;
        }
      }
      lds_barrier();
    }
#ifdef MVP_EMD_PROFILE
    const long long tpc = __builtin_readcyclecounter();
    if (t == 0) {
      s_prof[4] += (unsigned long long)nfall;
      s_prof[5] += (unsigned long long)(tpc - tpc0);
      s_prof[6] += (unsigned long long)(tpc0 - tpg);
    }
#endif
    if (t == 0) {
      s_nfall = 0;
      s_nmiss = 0;
      s_died = 0;  // (every wave has taken last round's compaction decision long ago)
    }
    lds_barrier();  // the round's bids are in the LDS arrays

    // ---------------- GetMax (emd_cuda.cu:181-194), exactly.  The bids of one object find
    // each other through the object's owner word: the last of them to write its number there
    // represents the group.
    const bool r_mine = t < U;
    int r_o = 0, r_j = -1, r_prev = 0xFFFF;
    float r_inc = 0.f;
    if (r_mine) {
      r_o = s_bo[t];
      r_j = s_bj[t];
      r_inc = s_binc[t];
      r_prev = s_owner[r_o];  // owner before this round (0xFFFF: free)
      s_gcnt[t] = 0;
      s_gmax[t] = 0u;
    }
    // (also: everything the previous round stored -- records and caches of the persons that
    // left the pool -- has landed before this round's Assign reads any of it)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // The cache of the person this bid evicts if it wins (nearly all do) is fetched now, so that
    // it travels while GetMax is resolved: 0-1 slots, 2-5 distances, 6 {tau, count}.
    v4u ch[7] = {};
    if (r_mine && !last && r_prev != 0xFFFF) {
      const unsigned pvc = off_cache + (unsigned)r_prev * kCacheRec;
#pragma unroll
      for (int c = 0; c < 7; ++c) ch[c] = ldg16(pvc + (unsigned)c * 16u);
    }
    if (r_mine) s_owner[r_o] = (unsigned short)(0x8000u | (unsigned)t);  // person ids are < 0x8000
    lds_barrier();
    int rep = 0;
    if (r_mine) {
      rep = s_owner[r_o] & 0x7FFF;
      atomicAdd(&s_gcnt[rep], 1);
      atomicMax(&s_gmax[rep], emd_f2ord(r_inc));
    }
    lds_barrier();
    bool r_win = true;
    if (r_mine && s_gcnt[rep] > 1) {  // rare: several bids on this object
      const float mxi = emd_ord2f(s_gmax[rep]);
      r_win = emd_in_band(r_inc, mxi);
      if (r_win)
        for (int v = 0; v < U; ++v)
          if (s_bo[v] == r_o && s_bj[v] > r_j && emd_in_band(s_binc[v], mxi)) r_win = false;
    }
    if (last) r_win = true;
    const bool r_evict = r_mine && r_win && !last && r_prev != 0xFFFF;
#ifdef MVP_EMD_PROFILE
    const long long tpr = __builtin_readcyclecounter();
#endif

    // ---------------- Assign (emd_cuda.cu:196-215), by the bid's own thread
    int a_e = 0;
    if (r_mine) {
      a_e = s_be[t];
      if (last) {
        st_i32(&ass[r_j], r_o);
      } else if (r_win) {
        if (r_prev == 0xFFFF) atomicAdd(&s_nfree, 1);  // a free object gets its first owner: one unassigned person fewer
        s_owner[r_o] = (unsigned short)r_j;
        s_price[r_o] = s_price[r_o] + r_inc;
        // the winner leaves the pool: its cache and hints go to scratch
        const unsigned myc = off_cache + (unsigned)r_j * kCacheRec;
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
        if (r_evict) st_i32(&ass[r_prev], -1);
        st_i32(&ass[r_j], r_o);
        // the entry: handed to the evicted person, or dead
        if (r_evict) {
          v4u *ecs = reinterpret_cast<v4u *>(&e_cs[a_e][0]);
          v4u *ecd = reinterpret_cast<v4u *>(&e_cd[a_e][0]);
          ecs[0] = ch[0];
          ecs[1] = ch[1];
          ecd[0] = ch[2];
          ecd[1] = ch[3];
          ecd[2] = ch[4];
          ecd[3] = ch[5];
          e_tau[a_e] = __uint_as_float(ch[6].x);
          e_i[a_e] = make_int4(r_prev, -1, -1, (int)ch[6].y);  // (the point is read where it is needed: phase C)
        } else {
          e_live[a_e] = 0;
          s_died = 1;
        }
      }
      // (lost: keeps its entry; the object's owner word still holds GetMax' marker, which the
      // winner's thread overwrites)
    }
    lds_barrier();
#ifdef MVP_EMD_PROFILE
    const long long tp2 = __builtin_readcyclecounter();
#endif
    // ---------------- next round's order list (entries are stable: it changes only when one died)
    if (s_died) {
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
        if (base != U0 - s_nfree) s_err = 1;  // cannot happen
      }
    }
    lds_barrier();
    }
    U = s_U;
#ifdef MVP_EMD_PROFILE
    const long long tp3 = __builtin_readcyclecounter();
    cyc_a += tpa - tp0;
    cyc_b += tpg - tpa;
    cyc_resolve += tpr - tpc;
    if (t == 0 && cloud < 2 && (it % 500) == 0)
      printf("solo cloud %d round %d: U %d | cumulative cycles A %lld B %lld wait %llu C %llu resolve %lld assign %lld compact %lld | misses %llu fallbacks %llu\n",
             cloud, it, U, cyc_a, cyc_b, s_prof[6], s_prof[5], cyc_resolve, cyc_assign, cyc_compact, s_prof[1], s_prof[4]);
    cyc_assign += tp2 - tpr;
    cyc_compact += tp3 - tp2;
#endif
  }
#ifdef MVP_EMD_PROFILE
  if (t == 0 && cloud < 2) {
    printf("solo cloud %d: rounds %lld bids %lld | cycles setup %lld phaseA %lld phaseB %lld resolve %lld assign %lld compact %lld\n",
           cloud, n_rounds, n_bids, cyc_setup, cyc_a, cyc_b, cyc_resolve, cyc_assign, cyc_compact);
    printf("solo cloud %d: hits %llu misses %llu (%llu cycles each, %llu entries scanned) fallbacks %llu (%llu cycles each); waiting for the slowest wave of phase B %llu\n",
           cloud, s_prof[0], s_prof[1], s_prof[2] / (s_prof[1] + 1), s_prof[3] / (s_prof[1] + 1), s_prof[4],
           s_prof[5] / (s_prof[4] + 1), s_prof[6]);
  }
#endif
  if (t == 0) {
    stats[0] = s_err ? -1 : stats[0] + n_rounds;
    atomicAdd(reinterpret_cast<unsigned long long *>(&stats[1]), (unsigned long long)n_bids);
  }
  // ---------------- CalcDist (emd_cuda.cu:217-226); slots -> object indices
  __syncthreads();
  for (int j = t; j < n; j += kEmdThreads) {
    const int s = ld_i32(&ass[j]);
    const float4 o = sc.obj[s];  // coordinates never change
    const float dx = xyz1[j * 3 + 0] - o.x;
    const float dy = xyz1[j * 3 + 1] - o.y;
    const float dz = xyz1[j * 3 + 2] - o.z;
    dist[j] = sqdist3(dx, dy, dz);
    ass[j] = sc.perm[s];
  }
}

// Cell records, lists, then the rounds.  `lists` = mvp_emd_scratch_bytes' list area.
void emd_solo_launch(int b, int n, const float *xyz1, float *dist, int *assignment, float eps, int iters,
                     char *scratch, char *lists, float delta, hipStream_t stream) {
  hipLaunchKernelGGL(emd_list_cells_kernel, dim3((kTailCells + 255) / 256, b), dim3(256), 0, stream, b, n, scratch, lists);
  hipLaunchKernelGGL(emd_list_persons_kernel, dim3(b), dim3(kEmdThreads), 0, stream, b, n, scratch, lists);
  hipLaunchKernelGGL(emd_list_build_kernel, dim3(kTailCells, b), dim3(kBuildWaves * kWave), 0, stream, b, n, scratch, lists);
#ifdef MVP_EMD_PROFILE
  if (const char *e = getenv("MVP_EMD_SOLO_STAGE"))
    if (atoi(e) == 1) return;  // timing aid: head + list build only (results are then incomplete)
#endif
  hipLaunchKernelGGL(emd_solo_kernel, dim3(b), dim3(kEmdThreads), 0, stream, b, n, xyz1, dist, assignment, eps, iters,
                     scratch, (const char *)lists, delta);
}

}  // namespace mvp
