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
//   * one workgroup owns the cloud: barriers are s_barrier, nothing is shared
//     with another CU, so there are no write-through stores, L1 bypasses or
//     all-gathers on the round's critical path;
//   * the PRICES (the only state a bid reads besides the static coordinates)
//     live in LDS (n <= 16384 floats), together with the cell index: per cell
//     one 16-byte record {fp16 bounding box rounded outwards, price lower
//     bound} -- one ds_read_b128 per cell test;
//   * the unassigned persons sit in a pool of kTailCap LDS entries {point,
//     hints, candidate cache}.  An entry is stable: a loser keeps it, a winner
//     hands it to the person it evicted (at most one), so the pool never grows
//     and needs no allocation; an order list of the live entries is rebuilt by
//     ballot each round;
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
//   * GetMax is resolved exactly in LDS (every bidder scans the round's <= 256
//     bids for its object: maximal increment, then the highest bidder inside
//     the reference's 1e-6 band, emd_cuda.cu:181-194) -- no atomics, no alarm
//     pass.
// A bid that misses its cache is one wave: cell enumeration from LDS, then ONE
// global round trip for the coordinates of the surviving cells' members (static
// data, plain cached loads), prices from LDS.
#include "emd_common.h"

namespace mvp {

constexpr int kTailCells = 1331;  // 11^3: n <= 16384 objects give g <= 11 (emd.hip: (g+1)^3 * 12 <= n)
constexpr int kStage = 64;        // candidates a search can stage for the cache (more: no cache this time)

// fp16 bounds of a float, rounded outwards (box lo down, box hi up), as bits
__device__ __forceinline__ unsigned half_bits_down(float x) {
  _Float16 h = (_Float16)x;  // round to nearest even; +-inf on overflow
  unsigned short b = __builtin_bit_cast(unsigned short, h);
  if ((float)h > x) {  // step to the next smaller half
    if (b == 0x0000u) b = 0x8001u;
    else if (b & 0x8000u) b += 1;
    else b -= 1;
  }
  return b;
}
__device__ __forceinline__ unsigned half_bits_up(float x) {
  _Float16 h = (_Float16)x;
  unsigned short b = __builtin_bit_cast(unsigned short, h);
  if ((float)h < x) {  // step to the next larger half
    if (b == 0x8000u) b = 0x0001u;
    else if (b & 0x8000u) b -= 1;
    else b += 1;
  }
  return b;
}
__device__ __forceinline__ float half_bits_to_float(unsigned b) {
  return (float)__builtin_bit_cast(_Float16, (unsigned short)b);
}

template <int CTRL>
__device__ __forceinline__ int quad_i32(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}

__global__ __launch_bounds__(kEmdThreads) void emd_tail_kernel(
    int b, int n, const float *__restrict__ xyz1, float *__restrict__ dist, int *assignment, float eps,
    int iters, char *scratch, float delta) {
  const int cloud = blockIdx.x;
  const size_t per_cloud = emd_scratch_per_cloud(n);
  char *cbase = scratch + (size_t)cloud * per_cloud;
  char *tail = scratch + (size_t)b * per_cloud;
  EmdResume *resume = reinterpret_cast<EmdResume *>(tail + (size_t)b * 256) + cloud;
  long long *stats = reinterpret_cast<long long *>(tail + (size_t)b * (256 + sizeof(EmdResume))) + 2 * (size_t)cloud;
  const int it0 = resume->next_it;
  if (it0 == 0) return;  // the cloud was finished by the first kernel (block-uniform)

  const int t = threadIdx.x;
  const int lane = t & (kWave - 1);
  const int wave = t >> 6;
  xyz1 += (size_t)cloud * n * 3;
  dist += (size_t)cloud * n;
  int *ass = assignment + (size_t)cloud * n;
  const EmdScratch sc = emd_carve(cbase, n);

  // Mutable global state (owners, assignment, person records, caches) is read
  // with L1-bypassing loads and written through: one workgroup is the only
  // reader and writer, but its waves must see each other's stores.
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(cbase, 0, (int)per_cloud, 0x00020000);
  const unsigned off_person = (unsigned)n * 32u;
  const unsigned off_cache = (unsigned)(reinterpret_cast<char *>(sc.cache) - cbase);
  auto ldg16 = [&](unsigned byte_off) -> v4u { return __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16); };
  auto stg16 = [&](v4u v, unsigned byte_off) { __builtin_amdgcn_raw_buffer_store_b128(v, rs, byte_off, 0, 16); };
  auto ld_i32 = [&](int *p) -> int { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto st_i32 = [&](int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };

  __shared__ float s_price[kTailMaxN];
  __shared__ uint4 s_cell[kTailCells + 1];  // {half2 x lo|hi, half2 y, half2 z, bits(price lower bound)}
  __shared__ int c_start[kTailCells + 1];
  // the pool of unassigned persons
  __shared__ float4 e_q[kTailCap];                              // the person's point
  __shared__ int4 e_i[kTailCap];                                // {person, hint slot 1, hint slot 2, cached candidates}
  __shared__ __attribute__((aligned(16))) unsigned short e_cs[kTailCap][kTailK];  // cached slots
  __shared__ __attribute__((aligned(16))) float e_cd[kTailCap][kTailK];           // their sqrtf distances
  __shared__ float e_tau[kTailCap];                             // >= value of every object NOT cached
  __shared__ unsigned char e_live[kTailCap];
  __shared__ unsigned short s_order[kTailCap];                  // live entries, ascending
  // this round's bids by list position
  __shared__ int s_bo[kTailCap], s_b2k[kTailCap], s_bj[kTailCap], s_be[kTailCap];
  __shared__ float s_binc[kTailCap];
  __shared__ unsigned short w_list[kEmdWaves][256];             // surviving cells of a search
  __shared__ int st_slot[kEmdWaves][kStage];                    // staged candidates of a search
  __shared__ float st_v[kEmdWaves][kStage], st_d[kEmdWaves][kStage];
  __shared__ int s_next, s_err, s_U, s_wcnt[4];

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
  for (int c = t; c <= ncell; c += kEmdThreads) c_start[c] = sc.cstart[c];
  for (int s = t; s < n; s += kEmdThreads) s_price[s] = sc.obj[s].w;
  if (t < kTailCap) {
    const bool live = t < U0;
    e_live[t] = live ? 1 : 0;
    s_order[t] = (unsigned short)t;
    if (live) {
      const int j = resume->list[t];
      const float4 lo = sc.person[2 * j];
      const float4 hi = sc.person[2 * j + 1];
      e_q[t] = lo;
      e_i[t] = make_int4(j, __float_as_int(hi.y), __float_as_int(hi.z), 0);  // caches start empty
    }
  }
  if (t == 0) {
    s_next = kEmdWaves;
    s_err = resume->pad;
    s_U = U0;
  }
  __syncthreads();
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

  // ------------------------------------------------------------ the rounds
  const int block_cnt = n / 1024;
  const bool caching = delta > 0.f;
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

    // ---------------- Bid (emd_cuda.cu:95-179): one wave per bidder, drawn from a counter
    int u = wave;
    for (int guard = 0; guard <= kTailCap && u < U; ++guard) {
      const int e = s_order[u];
      const float4 ra = e_q[e];
      const int4 rb = e_i[e];
      const int j = rb.x, p1 = rb.y, p2 = rb.z, cc = rb.w;
      const float qx = ra.x, qy = ra.y, qz = ra.z;
      float b1 = 0.f, b2 = 0.f, seed_b2 = -1e9f;
      int bk = -1, b2k = -1;
      bool hit = false;

      // (0) the cached candidates, re-evaluated at today's prices
      if (cc > 0) {
        float v = -1e9f;
        int slot = -1;
        if (lane < cc) {
          slot = e_cs[e][lane];
          v = emd_value_d(e_cd[e][lane], s_price[slot]);
        }
        float c1, c2;
        wave_top2(v, c1, c2);
        const float tau = e_tau[e];
        if (cc >= 2) seed_b2 = c2;  // two distinct real objects reach it: a bound of the final second best
        if (cc >= 2 && c2 >= tau && c1 > tau) {
          // every object outside the cache is worth <= tau: best and second best are in it
          hit = true;
          b1 = c1;
          b2 = c2;
          const unsigned long long m1 = __ballot(v == c1);
          const int l1 = __builtin_ctzll(m1);
          bk = __builtin_amdgcn_readlane(slot, l1);
          if (m1 & (m1 - 1)) {  // several best: the reference's order on ORIGINAL indices
            int bo = sc.perm[bk];
            unsigned long long mm = m1 & (m1 - 1);
            while (mm) {
              const int l = __builtin_ctzll(mm);
              mm &= mm - 1;
              const int kl = __builtin_amdgcn_readlane(slot, l);
              const int ol = sc.perm[kl];
              if (emd_precedes(ol, bo, n, tpu)) {
                b2k = bk;
                bk = kl;
                bo = ol;
              } else {
                b2k = kl;
              }
            }
          } else {
            const unsigned long long m2 = __ballot(v == c2) & ~(1ull << l1);
            b2k = m2 ? __builtin_amdgcn_readlane(slot, __builtin_ctzll(m2)) : -1;
          }
        }
      }

      if (!hit) {
        // (1) seed: without a usable cache, the second-largest exact value among
        // the home cell's members and the previous best / second best
        if (seed_b2 == -1e9f) {
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
        BidState st;
        st.b1 = -1e9f;
        st.b2 = -1e9f;
        st.bk = -1;
        st.b2k = -1;
        // the filter admits everything within delta of the second best (emd_common.h, kMargin,
        // with B2 := fl(b2 - delta)): what it skips is worth < b2 - delta
        st.tm = (3.0f - (seed_b2 - delta)) + kMargin;
        int nst = 0;  // staged candidates (wave-uniform)

        // evaluate one object per lane: filter, stage for the cache, fold what can matter
        auto consider = [&](bool valid, int s, const float4 &o) {
          const float p = valid ? s_price[s] : 0.f;
          const float sd = sqdist3(o.x - qx, o.y - qy, o.z - qz);
          const float tq = st.tm - p;
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
            const unsigned long long mf = __ballot(ps && v >= st.b2);
            if (mf) emd_fold(st, mf, v, s, n, tpu, sc.perm, delta);
          }
        };

        // (2) cells intersecting the cube |o - q|_inf <= tm (prices >= 0), 64 per step
        int ix0, iy0, iz0, nx, ny, nz;
        {
          const float r = st.tm * gg.invh + 1e-3f;  // slack covers index rounding
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
        const int sub = lane >> 4, sl = lane & 15;
        unsigned short *wl = w_list[wave];
        int nlist = 0;
        // (3) visit listed cells, 16 per step (a 16-lane row takes 4 cells): 4 independent
        // coordinate loads per lane in flight, prices from LDS
        auto visit = [&]() {
          for (int k0 = 0; k0 < nlist; k0 += 16) {
            int s[4], s1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int k = k0 + r * 4 + sub;
              s[r] = 0;
              s1[r] = 0;
              if (k < nlist) {
                const int cc2 = wl[k];
                s[r] = c_start[cc2] + sl;
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
              bool mine = false;  // cells with more than 16 members (rare): next 16
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                s[r] += 16;
                mine |= s[r] < s1[r];
              }
              more = __any(mine);
            }
          }
          nlist = 0;
        };
        // a search cube covering most of the grid: scan the cell-sorted objects linearly
        const bool linear = 2 * nsub > ncell;
        if (linear) {
          for (int base = 0; base < n; base += 4 * kWave) {  // n % 1024 == 0
            float4 o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = sc.obj[base + r * kWave + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) consider(true, base + r * kWave + lane, o[r]);
          }
        }
        for (int cb = 0; cb < (linear ? 0 : nsub); cb += kWave) {
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
            const float tq = st.tm - __uint_as_float(cr.w);
            cpass = tq >= 0.f && sqdist3(dx, dy, dz) <= tq * tq;  // empty cell: inf
          }
          const unsigned long long cmask = __ballot(cpass);
          if (cpass) wl[nlist + __builtin_popcountll(cmask & ((1ull << lane) - 1ull))] = (unsigned short)c;
          nlist += __builtin_popcountll(cmask);
          if (nlist > 256 - kWave) visit();  // keep room for the next 64
        }
        visit();

        if (st.bk < 0) {  // cannot happen (>= 2 objects always survive); never index with -1
          if (lane == 0) s_err = 1;
          st.bk = 0;
          st.b2k = -1;
        }
        bk = st.bk;
        b2k = st.b2k;
        b1 = st.b1;
        b2 = st.b2;

        // (4) rebuild the cache from the staged candidates
        if (caching) {
          float tau = st.b2 - delta;  // everything the search skipped is worth less
          int newcc = 0;
          if (nst <= kStage) {
            const bool have = lane < nst;
            const float v = have ? st_v[wave][lane] : -1e9f;
            const int slot = have ? st_slot[wave][lane] : 0;
            const float d = have ? st_d[wave][lane] : 0.f;
            const bool keep = have && v >= tau;
            const unsigned long long km = __ballot(keep);
            const int kc = __builtin_popcountll(km);
            int rank = 0;  // position in descending value order (ties: lower lane first)
            unsigned long long mm = km;
            while (mm) {
              const int l = __builtin_ctzll(mm);
              mm &= mm - 1;
              const float vl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
              rank += (vl > v || (vl == v && l < lane)) ? 1 : 0;
            }
            if (kc > kTailK) {  // the best kTailK stay; the next one bounds the rest
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
      }

      int drawn = 0;
      if (lane == 0) {
        s_bo[u] = bk;
        s_b2k[u] = b2k;
        s_binc[u] = b1 - b2 + eps;
        s_bj[u] = j;
        s_be[u] = e;
        drawn = atomicAdd(&s_next, 1);
      }
      u = __builtin_amdgcn_readlane(drawn, 0);
    }
    __syncthreads();

    // ---------------- GetMax + Assign (emd_cuda.cu:181-215): four threads per bidder
    {
      const int ub = t >> 2, q4 = t & 3;
      const bool act = ub < U;
      int o = -1, j = -1, e = 0;
      float inc = 0.f;
      if (act) {
        o = s_bo[ub];
        inc = s_binc[ub];
        j = s_bj[ub];
        e = s_be[ub];
      }
      // maximal increment bid on my object, then: am I the highest bidder inside its band?
      float mi = -1e9f;
      for (int v = q4; v < U; v += 4)
        if (s_bo[v] == o) mi = __builtin_fmaxf(mi, s_binc[v]);
      mi = __builtin_fmaxf(mi, __int_as_float(quad_i32<0xB1>(__float_as_int(mi))));
      mi = __builtin_fmaxf(mi, __int_as_float(quad_i32<0x4E>(__float_as_int(mi))));
      int beaten = 0;
      for (int v = q4; v < U; v += 4)
        if (s_bo[v] == o && s_bj[v] > j && emd_in_band(s_binc[v], mi)) beaten = 1;
      beaten |= quad_i32<0xB1>(beaten);
      beaten |= quad_i32<0x4E>(beaten);
      const bool win = act && (last || (emd_in_band(inc, mi) && !beaten));

      if (win && last) {
        if (q4 == 0) st_i32(&ass[j], o);
      } else if (win) {
        int prev = -1;
        float4 oo = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q4 == 0) {
          prev = ld_i32(&sc.ostate[o].z);
          oo = sc.obj[o];  // coordinates only (static)
        }
        prev = quad_i32<0x00>(prev);  // quad_perm [0,0,0,0]
        const bool evict = prev != -1;
        // the winner leaves the pool: its cache goes to scratch (chunks of 16 bytes: 0-1 slots,
        // 2-5 distances, 6 {tau, count}); the person it evicts takes the entry over
        const unsigned myc = off_cache + (unsigned)j * kCacheRec;
        const unsigned pvc = off_cache + (unsigned)prev * kCacheRec;
        v4u in0 = {0u, 0u, 0u, 0u}, in1 = {0u, 0u, 0u, 0u};
        if (caching) {
          if (evict) {
            in0 = ldg16(pvc + (2u * q4) * 16u);
            if (q4 < 3) in1 = ldg16(pvc + (2u * q4 + 1u) * 16u);
          }
          const int cc = e_i[e].w;
          if (q4 == 0) {
            const v4u *src = reinterpret_cast<const v4u *>(&e_cs[e][0]);
            stg16(src[0], myc);
            stg16(src[1], myc + 16u);
          } else if (q4 < 3) {
            const v4u *src = reinterpret_cast<const v4u *>(&e_cd[e][0]);
            stg16(src[2 * q4 - 2], myc + (2u * q4) * 16u);
            stg16(src[2 * q4 - 1], myc + (2u * q4 + 1u) * 16u);
          } else {
            v4u r;
            r.x = __float_as_uint(e_tau[e]);
            r.y = (unsigned)cc;
            r.z = 0u;
            r.w = 0u;
            stg16(r, myc + 96u);
          }
        }
        float4 plo = make_float4(0.f, 0.f, 0.f, 0.f);
        int h1 = -1, h2 = -1;
        if (q4 == 0) {
          if (evict) {
            const v4u a = ldg16(off_person + (2u * (unsigned)prev) * 16u);
            const v4u bb = ldg16(off_person + (2u * (unsigned)prev + 1u) * 16u);
            plo = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), 0.f);
            h1 = (int)bb.y;
            h2 = (int)bb.z;
            st_i32(&ass[prev], -1);
          }
          st_i32(&sc.ostate[o].z, j);
          st_i32(&ass[j], o);
          v4u hi;
          hi.x = (unsigned)o;
          hi.y = (unsigned)o;
          hi.z = (unsigned)s_b2k[ub];
          hi.w = __float_as_uint(inc);
          stg16(hi, off_person + (2u * (unsigned)j + 1u) * 16u);
          // the price; the cell's lower bound only when (one of) its cheapest got dearer
          const float pold = s_price[o];
          const float pnew = pold + inc;
          s_price[o] = pnew;
          const int c = emd_cell(gg, oo.x, oo.y, oo.z);
          float *lbp = reinterpret_cast<float *>(&s_cell[c]) + 3;
          if (pold <= *lbp) {
            float pm = pnew;
            const int e1 = c_start[c + 1];
            for (int s = c_start[c]; s < e1; ++s) pm = __builtin_fminf(pm, s == o ? pm : s_price[s]);
            *lbp = pm;
          }
        }
        // the entry: handed to the evicted person, or dead
        if (evict) {
          if (caching) {
            if (q4 == 0) {
              v4u *dst = reinterpret_cast<v4u *>(&e_cs[e][0]);
              dst[0] = in0;
              dst[1] = in1;
            } else if (q4 < 3) {
              v4u *dst = reinterpret_cast<v4u *>(&e_cd[e][0]);
              dst[2 * q4 - 2] = in0;
              dst[2 * q4 - 1] = in1;
            } else {
              e_tau[e] = __uint_as_float(in0.x);
              e_i[e].w = (int)in0.y;
            }
          }
          if (q4 == 0) {
            e_q[e] = plo;
            e_i[e].x = prev;
            e_i[e].y = h1;
            e_i[e].z = h2;
            if (!caching) e_i[e].w = 0;
          }
        } else if (q4 == 0) {
          e_live[e] = 0;
        }
      } else if (act && q4 == 0) {
        // lost: keeps its entry; this bid's best / second best seed the next one
        e_i[e].y = o;
        e_i[e].z = s_b2k[ub];
      }
    }
    __syncthreads();

    // ---------------- next round's order list: the live entries, by ballot
    unsigned long long lm = 0ull;
    bool live = false;
    if (t < kTailCap) {
      live = e_live[t] != 0;
      lm = __ballot(live);
      if (lane == 0) s_wcnt[wave] = __builtin_popcountll(lm);
    }
    __syncthreads();
    if (t < kTailCap) {
      int base = 0;
      for (int w = 0; w < wave; ++w) base += s_wcnt[w];
      if (live) s_order[base + __builtin_popcountll(lm & ((1ull << lane) - 1ull))] = (unsigned short)t;
    }
    if (t == 0) {
      s_U = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
      s_next = kEmdWaves;
    }
    __syncthreads();
    U = s_U;
  }

  if (t == 0) {
    stats[0] = s_err ? -1 : stats[0] + n_rounds;
    stats[1] += n_bids;
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

void emd_tail_launch(int b, int n, const float *xyz1, float *dist, int *assignment, float eps, int iters,
                     char *scratch, float delta, hipStream_t stream) {
  hipLaunchKernelGGL(emd_tail_kernel, dim3(b), dim3(kEmdThreads), 0, stream, b, n, xyz1, dist, assignment, eps,
                     iters, scratch, delta);
}

}  // namespace mvp
