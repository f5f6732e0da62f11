// The spatial index of the EMD auction's pruned search (round 6; replaces the uniform 12^3-cell grid of rounds 1-5).
//
// The reference's Bid scans all n objects per bidder and round (utils/metrics/EMD/emd_cuda.cu:95-179); the kernels here
// evaluate only objects that can still change {best, second best, best index} (emd_common.h: kMargin) and find them
// through boxes.  A fixed grid sized for ~12 objects per cell of a UNIFORM VOLUME holds 30-200 per occupied cell on the
// 2-manifolds MVP clouds are samples of (completion/dataset.py:21-34): a search loaded 23-33 chunks of 16 objects to find
// the 3-5 that pass (profiles/r5_emd_surfaces.txt, r6_emd_index_study.txt).  So the index follows the cloud instead:
//   * objects are sorted along a HILBERT curve (9 bits per axis over the cube that bounds both clouds);
//   * a LEAF is 16 consecutive slots of that order (n <= 16384; 16 * 2^k slots above, so that there are at most 1024
//     leaves) -- 16 objects whatever the cloud's intrinsic dimension --, a NODE is 16 consecutive leaves (at most 64: one
//     lane each);
//   * per leaf and per node, in LDS: the exact bounding box of the members and a lower bound of their prices (prices only
//     rise, so a stale bound stays valid; a winner's leaf is re-scanned).
// A search tests the <= 64 node boxes in one step, the 16 leaves of every passing node (four nodes per step) and visits
// the passing leaves, 16 per memory round trip.  Every test is the same conservative one as before (distance to the box
// and price bound against the running threshold), so the result stays bit-identical to the exhaustive scan.
// Counted on the exact CPU auction before anything was built (tools/emd_index_study.c, 8 x 16384 points, rounds 300-800):
// chunks loaded per search, grid -> leaves: uniform volume 8.8 -> 9.8, sphere gt + noise 13.4 -> 7.0, sphere independent
// samples 14.3 -> 9.6, chair gt + noise 22.6 -> 5.7, chair independent 33.0 -> 10.1 (Morton order: 16.4 on the uniform
// volume -- the curve matters).
#pragma once
#include "emd_common.h"

namespace mvp {

constexpr int kMaxLeaves = 1024;
constexpr int kMaxNodes = 64;     // one lane each
constexpr int kNodeFan = 16;      // leaves per node: one 16-lane row each
constexpr int kHilbertBits = 9;   // per axis: 27-bit keys, three 9-bit digits
constexpr int kSortDigits = 1 << kHilbertBits;
#ifndef MVP_EMD_VISIT_LOADS
#define MVP_EMD_VISIT_LOADS 4
#endif
#ifndef MVP_EMD_ROW_VISIT_LOADS
#define MVP_EMD_ROW_VISIT_LOADS 4
#endif
constexpr int kRowVisitLoads = MVP_EMD_ROW_VISIT_LOADS;   // ... and per step of a four-bidders-per-wave search (emd.hip)
constexpr int kVisitLoads = MVP_EMD_VISIT_LOADS;   // leaves a 16-lane row loads per visit step of a one-bidder-per-wave search (4 rows: 16 leaves per step)
static_assert(kMaxNodes * kNodeFan == kMaxLeaves && kMaxNodes == kWave, "a node per lane, a leaf per lane of a row");

// log2 of the slots of a leaf: 4 (16 slots) up to 16384 points, then as many more as keep the leaves at <= 1024.
// n % 1024 == 0 and n <= 2^20, so a leaf holds 16 .. 1024 slots and n is a multiple of it.
__host__ __device__ inline int emd_leaf_shift(int n) {
  int s = 4;
  while ((n >> s) > kMaxLeaves) ++s;
  return s;
}

// Hilbert index of a point of the 512^3 lattice (Skilling's transpose form, unrolled for three axes): consecutive
// indices are face-adjacent lattice points, so a run of the sorted order is a compact piece of the cloud.
__device__ __forceinline__ unsigned emd_hilbert(unsigned x, unsigned y, unsigned z) {
  constexpr unsigned M = 1u << (kHilbertBits - 1);
#pragma unroll
  for (unsigned Q = M; Q > 1; Q >>= 1) {
    const unsigned P = Q - 1;
    if (x & Q) x ^= P;   // (axis 0 against itself: invert)
    if (y & Q) x ^= P; else { const unsigned t = (x ^ y) & P; x ^= t; y ^= t; }
    if (z & Q) x ^= P; else { const unsigned t = (x ^ z) & P; x ^= t; z ^= t; }
  }
  y ^= x;
  z ^= y;
  unsigned t = 0;
#pragma unroll
  for (unsigned Q = M; Q > 1; Q >>= 1)
    if (z & Q) t ^= Q - 1;
  x ^= t; y ^= t; z ^= t;
  unsigned h = 0;
#pragma unroll
  for (int b = kHilbertBits - 1; b >= 0; --b) h = (h << 3) | (((x >> b) & 1u) << 2) | (((y >> b) & 1u) << 1) | ((z >> b) & 1u);
  return h;
}

// Lattice frame of a cloud pair: the cube of side `ext` at (lox, loy, loz) that bounds both clouds.
struct HilbertFrame {
  float lox, loy, loz, scale;   // scale = 512 / ext
};
__device__ __forceinline__ unsigned emd_hilbert_key(const HilbertFrame &f, float x, float y, float z) {
  constexpr int kTop = (1 << kHilbertBits) - 1;
  const unsigned ix = (unsigned)min(kTop, max(0, (int)((x - f.lox) * f.scale)));
  const unsigned iy = (unsigned)min(kTop, max(0, (int)((y - f.loy) * f.scale)));
  const unsigned iz = (unsigned)min(kTop, max(0, (int)((z - f.loz) * f.scale)));
  return emd_hilbert(ix, iy, iz);
}

// One pass of a stable least-significant-digit radix sort of n 64-bit entries {key 32 | payload 32} by the 9-bit digit at
// `shift` of the key, on the calling workgroup (1024 threads; n % 1024 == 0): wave w owns the w-th sixteenth of the
// source, counts its digits, and after a prefix sum over (digit, wave) scatters its entries in order -- the rank of an
// entry among the wave's current 64 with the same digit comes from nine ballots.  hist: 512 x 16 ints of LDS.
__device__ __forceinline__ void emd_sort_pass(const u64 *__restrict__ src, u64 *__restrict__ dst, int n, int shift,
                                              int *hist, int *s_wsum) {
  const int t = threadIdx.x, lane = t & (kWave - 1), wave = t >> 6;
  const int seg = n / kEmdWaves;   // a multiple of 64
  for (int e = t; e < kSortDigits * kEmdWaves; e += kEmdThreads) hist[e] = 0;
  __syncthreads();
  // (eight entries per lane are loaded before the first is used: the passes are chains of dependent memory round trips
  // otherwise -- 16 per wave and phase at 16384 points)
  for (int i0 = 0; i0 < seg; i0 += 8 * kWave) {
    u64 e[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = i0 + k * kWave + lane < seg ? src[(size_t)wave * seg + i0 + k * kWave + lane] : 0ull;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (i0 + k * kWave < seg) atomicAdd(&hist[((unsigned)(e[k] >> (32 + shift)) & (kSortDigits - 1)) * kEmdWaves + wave], 1);
  }
  __syncthreads();
  {  // exclusive prefix sum over the 8192 counters in (digit, wave) order: 8 consecutive ones per thread
    int v[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k] = hist[8 * t + k];
      sum += v[k];
    }
    int incl = sum;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int o = __shfl_up(incl, off, kWave);
      if (lane >= off) incl += o;
    }
    if (lane == kWave - 1) s_wsum[wave] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int w = 0; w < wave; ++w) base += s_wsum[w];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      hist[8 * t + k] = base;
      base += v[k];
    }
  }
  __syncthreads();
  for (int i0 = 0; i0 < seg; i0 += 8 * kWave) {
    u64 ev[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) ev[k] = i0 + k * kWave + lane < seg ? src[(size_t)wave * seg + i0 + k * kWave + lane] : 0ull;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (i0 + k * kWave >= seg) break;   // (seg is a multiple of 64: whole batches)
      const u64 e = ev[k];
      const unsigned d = (unsigned)(e >> (32 + shift)) & (kSortDigits - 1);
      unsigned long long peers = ~0ull;
#pragma unroll
      for (int bit = 0; bit < kHilbertBits; ++bit) {
        const unsigned long long bm = __ballot((d >> bit) & 1u);
        peers &= ((d >> bit) & 1u) ? bm : ~bm;
      }
      const int rank = __builtin_popcountll(peers & ((1ull << lane) - 1ull));
      int *cnt = &hist[d * kEmdWaves + wave];
      const int base = *cnt;
      dst[base + rank] = e;
      // (every peer has read the counter -- LDS operations of a wave are in order --; the last of them moves it on)
      if (lane == 63 - (int)__builtin_clzll(peers)) *cnt = base + __builtin_popcountll(peers);
    }
  }
  __syncthreads();
}

// min / max over each 16-lane row, valid in every lane of the row
__device__ __forceinline__ float emd_row_min(float v) {
  const float inf = __builtin_inff();
  v = __builtin_fminf(v, dpp_f32<0xB1, 0xF>(inf, v));    // quad_perm [1,0,3,2]
  v = __builtin_fminf(v, dpp_f32<0x4E, 0xF>(inf, v));    // quad_perm [2,3,0,1]
  v = __builtin_fminf(v, dpp_f32<0x141, 0xF>(inf, v));   // row_half_mirror
  v = __builtin_fminf(v, dpp_f32<0x140, 0xF>(inf, v));   // row_mirror
  return v;
}
__device__ __forceinline__ float emd_row_max(float v) { return -emd_row_min(-v); }

// The leaves' and nodes' exact boxes and exact price minima from the sorted objects as they are in memory (16-lane row
// per leaf), into the calling workgroup's LDS: l_lo / l_hi [kMaxLeaves], n_lo / n_hi [kMaxNodes]; entries behind the
// cloud's last leaf / node are empty boxes (every test fails on them).  Ends with a workgroup barrier.
//   lo = {min x, y, z, price lower bound}, hi = {max x, y, z, -}
__device__ __forceinline__ void emd_index_boxes(float4 *l_lo, float4 *l_hi, float4 *n_lo, float4 *n_hi,
                                                const float4 *__restrict__ obj, int n, int lshift) {
  const int t = threadIdx.x, sl = t & 15;
  const int nleaf = n >> lshift;
  const float inf = __builtin_inff();
  auto reduce_store = [&](int leaf, float lx, float ly, float lz, float pm, float hx, float hy, float hz) {
    lx = emd_row_min(lx); ly = emd_row_min(ly); lz = emd_row_min(lz); pm = emd_row_min(pm);
    hx = emd_row_max(hx); hy = emd_row_max(hy); hz = emd_row_max(hz);
    if (sl == 0) {
      l_lo[leaf] = make_float4(lx, ly, lz, leaf < nleaf ? pm : 0.f);
      l_hi[leaf] = make_float4(hx, hy, hz, 0.f);
    }
  };
  if (lshift == 4) {
    // (16 slots per leaf, up to 16384 points: four leaves' loads in flight per row and iteration)
    for (int l0 = t >> 4; l0 < kMaxLeaves; l0 += 4 * (kEmdThreads / 16)) {
      float4 o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int leaf = l0 + j * (kEmdThreads / 16);
        o[j] = leaf < nleaf ? obj[leaf * 16 + sl] : make_float4(inf, inf, inf, inf);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int leaf = l0 + j * (kEmdThreads / 16);
        const bool in = leaf < nleaf;
        reduce_store(leaf, o[j].x, o[j].y, o[j].z, o[j].w, in ? o[j].x : -inf, in ? o[j].y : -inf, in ? o[j].z : -inf);
      }
    }
  } else {
    for (int leaf = t >> 4; leaf < kMaxLeaves; leaf += kEmdThreads / 16) {
      float lx = inf, ly = inf, lz = inf, pm = inf, hx = -inf, hy = -inf, hz = -inf;
      if (leaf < nleaf) {
        for (int s = (leaf << lshift) + sl; s < ((leaf + 1) << lshift); s += 16) {
          const float4 o = obj[s];
          lx = __builtin_fminf(lx, o.x); ly = __builtin_fminf(ly, o.y); lz = __builtin_fminf(lz, o.z);
          hx = __builtin_fmaxf(hx, o.x); hy = __builtin_fmaxf(hy, o.y); hz = __builtin_fmaxf(hz, o.z);
          pm = __builtin_fminf(pm, o.w);
        }
      }
      reduce_store(leaf, lx, ly, lz, pm, hx, hy, hz);
    }
  }
  __syncthreads();
  if (t < kMaxNodes) {
    float lx = inf, ly = inf, lz = inf, pm = inf, hx = -inf, hy = -inf, hz = -inf;
    for (int k = 0; k < kNodeFan; ++k) {
      const int leaf = t * kNodeFan + k;
      if (leaf < nleaf) {
        const float4 a = l_lo[leaf], c = l_hi[leaf];
        lx = __builtin_fminf(lx, a.x); ly = __builtin_fminf(ly, a.y); lz = __builtin_fminf(lz, a.z); pm = __builtin_fminf(pm, a.w);
        hx = __builtin_fmaxf(hx, c.x); hy = __builtin_fmaxf(hy, c.y); hz = __builtin_fmaxf(hz, c.z);
      }
    }
    n_lo[t] = make_float4(lx, ly, lz, t * kNodeFan < nleaf ? pm : 0.f);
    n_hi[t] = make_float4(hx, hy, hz, 0.f);
  }
  __syncthreads();
}

// A node's price bound from its leaves' current bounds (LDS only; whatever mix of old and new leaf bounds it reads is a
// valid lower bound).  Called by one lane per node once a round, off the critical path.
__device__ __forceinline__ void emd_node_price(const float4 *l_lo, float4 *n_lo, int node, int nleaf) {
  if (node * kNodeFan >= nleaf) return;
  float pm = __builtin_inff();
#pragma unroll
  for (int k = 0; k < kNodeFan; ++k)
    if (node * kNodeFan + k < nleaf) pm = __builtin_fminf(pm, l_lo[node * kNodeFan + k].w);
  n_lo[node].w = pm;
}

// The conservative box test of a search (emd_common.h: kMargin): can the box {lo.xyz .. hi.xyz} with price bound lo.w
// hold an object whose value reaches the running second best?  tm = fl(fl(3 - B2) + kMargin).  Empty box: false.
__device__ __forceinline__ bool emd_box_pass(const float4 lo, const float4 hi, float qx, float qy, float qz, float tm) {
  const float dx = __builtin_fmaxf(__builtin_fmaxf(lo.x - qx, qx - hi.x), 0.f);
  const float dy = __builtin_fmaxf(__builtin_fmaxf(lo.y - qy, qy - hi.y), 0.f);
  const float dz = __builtin_fmaxf(__builtin_fmaxf(lo.z - qz, qz - hi.z), 0.f);
  const float tq = tm - lo.w;
  return tq >= 0.f && sqdist3(dx, dy, dz) <= tq * tq;
}

}  // namespace mvp
