// Chamfer distance for gfx950: bidirectional nearest-neighbour squared
// distance + arg-min (forward) and its gradient (backward).
//
// Replaces NmDistanceKernel / NmDistanceGradKernel of the reference
// (utils/metrics/CD/chamfer3D/chamfer3D.cu:12-134,155-174).  Not a port: the
// reference stages 512-point tiles and tracks (best, best_i) per candidate in
// a fixed 32x16 grid.  Here
//   * both directions run in ONE launch (blockIdx.z), sized to the problem so
//     every cloud/query block gets its own workgroup (>> 256 workgroups);
//   * each lane owns Q query points in registers and streams candidate points
//     from an LDS tile with wave-uniform ds_read_b128 (broadcast, no bank
//     conflicts): 3 reads deliver 4 candidates;
//   * the hot loop only tracks the running MINIMUM over sub-tiles of 16
//     candidates (v_min), remembering which sub-tile last lowered it; the
//     arg-min is recovered afterwards by re-scanning that single sub-tile
//     ("min-then-locate").  This removes the compare+2 selects per pair that
//     index tracking costs, and keeps the reference's tie rule exactly:
//     strict `<` between sub-tiles (earliest sub-tile wins) and first equal
//     candidate inside it (lowest index wins) -- chamfer3D.cu:36,46,126.
//   * arithmetic is the canonical fma chain shared with the CPU oracle.
#include "common.h"
#include "cs_sort.h"

namespace mvp {

constexpr int kCdThreads = 256;
constexpr int kCdTile = 1024;  // candidates per LDS tile (12 KiB)
constexpr int kCdSub = 16;     // candidates per min-tracking sub-tile

template <int Q>
__global__ __launch_bounds__(kCdThreads) void nm_distance_kernel(
    int n1, int n2, const float *__restrict__ xyz1,
    const float *__restrict__ xyz2, float *__restrict__ dist1,
    float *__restrict__ dist2, int *__restrict__ idx1, int *__restrict__ idx2) {
  // Direction 0: queries = xyz1 (n1 points), candidates = xyz2 (n2 points).
  // Direction 1: roles swapped.
  const int dir = blockIdx.z;
  const int n = dir == 0 ? n1 : n2;
  const int m = dir == 0 ? n2 : n1;
  if ((int)blockIdx.x * (kCdThreads * Q) >= n) return;
  const int cloud = blockIdx.y;
  const float *__restrict__ qpts = (dir == 0 ? xyz1 : xyz2) + (size_t)cloud * n * 3;
  const float *__restrict__ cpts = (dir == 0 ? xyz2 : xyz1) + (size_t)cloud * m * 3;
  float *__restrict__ result = (dir == 0 ? dist1 : dist2) + (size_t)cloud * n;
  int *__restrict__ result_i = (dir == 0 ? idx1 : idx2) + (size_t)cloud * n;

  __shared__ __attribute__((aligned(16))) float tile[kCdTile * 3];

  const int tid = threadIdx.x;
  float qx[Q], qy[Q], qz[Q], best[Q];
  int bsub[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    int j = blockIdx.x * (kCdThreads * Q) + q * kCdThreads + tid;
    j = j < n ? j : n - 1;
    qx[q] = qpts[j * 3 + 0];
    qy[q] = qpts[j * 3 + 1];
    qz[q] = qpts[j * 3 + 2];
    best[q] = __builtin_inff();
    bsub[q] = 0;
  }

  for (int k2 = 0; k2 < m; k2 += kCdTile) {
    const int cnt = min(kCdTile, m - k2);
    // Stage the tile (AoS, exactly as in global memory); pad the tail of the
    // last sub-tile with +inf so padded candidates evaluate to d = +inf.
    const int padded = ((cnt + kCdSub - 1) / kCdSub) * kCdSub;
    for (int t = tid; t < padded * 3; t += kCdThreads)
      tile[t] = t < cnt * 3 ? cpts[(size_t)k2 * 3 + t] : __builtin_inff();
    __syncthreads();

    const float4 *__restrict__ t4 = reinterpret_cast<const float4 *>(tile);
    const int nsub = padded / kCdSub;
    for (int s = 0; s < nsub; ++s) {
      float tmin[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) tmin[q] = __builtin_inff();
#pragma unroll
      for (int g = 0; g < kCdSub / 4; ++g) {
        const float4 a = t4[(s * (kCdSub / 4) + g) * 3 + 0];
        const float4 b = t4[(s * (kCdSub / 4) + g) * 3 + 1];
        const float4 c = t4[(s * (kCdSub / 4) + g) * 3 + 2];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const float d0 = sqdist3(a.x - qx[q], a.y - qy[q], a.z - qz[q]);
          const float d1 = sqdist3(a.w - qx[q], b.x - qy[q], b.y - qz[q]);
          const float d2 = sqdist3(b.z - qx[q], b.w - qy[q], c.x - qz[q]);
          const float d3 = sqdist3(c.y - qx[q], c.z - qy[q], c.w - qz[q]);
          tmin[q] = __builtin_fminf(__builtin_fminf(tmin[q], d0), d1);
          tmin[q] = __builtin_fminf(__builtin_fminf(tmin[q], d2), d3);
        }
      }
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const bool lower = tmin[q] < best[q];
        best[q] = lower ? tmin[q] : best[q];
        bsub[q] = lower ? (k2 / kCdSub + s) : bsub[q];
      }
    }
    __syncthreads();
  }

  // Locate: first candidate of the winning sub-tile whose distance equals the
  // minimum (bitwise-identical recomputation).
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int j = blockIdx.x * (kCdThreads * Q) + q * kCdThreads + tid;
    if (j >= n) continue;
    const int k0 = bsub[q] * kCdSub;
    const int k1 = min(k0 + kCdSub, m);
    int bi = k0;
    bool found = false;
    for (int k = k0; k < k1; ++k) {
      const float d = sqdist3(cpts[k * 3 + 0] - qx[q], cpts[k * 3 + 1] - qy[q],
                              cpts[k * 3 + 2] - qz[q]);
      if (!found && d == best[q]) {
        bi = k;
        found = true;
      }
    }
    result[j] = best[q];
    result_i[j] = bi;
  }
}

// ---------------------------------------------------------------------------
// Spatially sorted variant for large clouds: the same result with most of the
// B*N*M pairs never evaluated.
//
// (1) chamfer_sort_kernel buckets each cloud side into 16^3 Morton-ordered
//     cells (counting sort in LDS) and writes, into caller-provided scratch,
//     the points in that order as float4 {x, y, z, bits(original index)}, the
//     bounding box of every run of 16 sorted points (a "tile") and of every run
//     of 1024 (a "batch").
// (2) nm_distance_sorted_kernel keeps the brute-force skeleton -- a lane owns Q
//     queries, candidates stream through LDS batches -- but both sides are in
//     Morton order, so the 64*Q queries of a wave sit in a small box and a
//     tile / batch is SKIPPED when the distance between that box and the
//     tile's box already exceeds the worst current best of the wave.  Batches
//     are visited outwards from the query block's own position in the order,
//     so the bests are tight after the first batch or two.
// Exactness: the box distance uses the same subtract / fma chain as sqdist3 on
// per-axis gaps that are <= every member pair's |difference| (float subtract,
// multiply and fma are monotone), so it never exceeds a member pair's computed
// distance, and a tile is skipped only on strict `>`: a candidate that equals
// the current best is still evaluated.  Ties go to the lowest ORIGINAL index
// (chamfer3D.cu:36,46,126) by minimising the key {distance bits, original
// index} -- the visiting order no longer is the index order.
__device__ __forceinline__ int cs_spread4(int v) {  // bit i -> bit 3i
  v &= 0xF;
  v = (v | (v << 4)) & 0xC3;
  v = (v | (v << 2)) & 0x249;
  return v;
}

__global__ __launch_bounds__(kCsThreads) void chamfer_sort_kernel(
    int n1, int n2, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
    char *__restrict__ scratch) {
  const int side = blockIdx.x, cloud = blockIdx.y;
  const int cnt = side == 0 ? n1 : n2;
  const long long per_cloud = cs_side_bytes(n1) + cs_side_bytes(n2);
  const CsSide out = cs_carve(scratch + (size_t)cloud * per_cloud + (side ? cs_side_bytes(n1) : 0), cnt);
  const float *__restrict__ in = (side == 0 ? xyz1 : xyz2) + (size_t)cloud * cnt * 3;
  const int cp = (int)cs_round_up(cnt);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

  __shared__ int s_cnt[kCsCells];
  __shared__ int s_start[kCsCells];
  __shared__ float s_red[6][kCsThreads / 64];
  __shared__ int s_wsum[kCsThreads / 64];

  // bounding box
  float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
  float mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  for (int k = t; k < cnt; k += kCsThreads) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = in[k * 3 + a];
      mn[a] = __builtin_fminf(mn[a], v);
      mx[a] = __builtin_fmaxf(mx[a], v);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      mn[a] = __builtin_fminf(mn[a], __shfl_xor(mn[a], off, 64));
      mx[a] = __builtin_fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
    }
    if (lane == 0) {
      s_red[a][wave] = mn[a];
      s_red[3 + a][wave] = mx[a];
    }
  }
  for (int c = t; c < kCsCells; c += kCsThreads) s_cnt[c] = 0;
  __syncthreads();
  float lo[3], ext = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float l = s_red[a][0], h = s_red[3 + a][0];
    for (int w = 1; w < kCsThreads / 64; ++w) {
      l = __builtin_fminf(l, s_red[a][w]);
      h = __builtin_fmaxf(h, s_red[3 + a][w]);
    }
    lo[a] = l;
    ext = __builtin_fmaxf(ext, h - l);
  }
  if (!(ext > 0.f) || !(ext < 3.0e38f)) ext = 1.f;
  const float invh = 16.f / ext;
  auto cell_of = [&](float x, float y, float z) {
    const int ix = min(15, max(0, (int)((x - lo[0]) * invh)));
    const int iy = min(15, max(0, (int)((y - lo[1]) * invh)));
    const int iz = min(15, max(0, (int)((z - lo[2]) * invh)));
    return cs_spread4(ix) | (cs_spread4(iy) << 1) | (cs_spread4(iz) << 2);
  };
  for (int k = t; k < cnt; k += kCsThreads)
    atomicAdd(&s_cnt[cell_of(in[k * 3 + 0], in[k * 3 + 1], in[k * 3 + 2])], 1);
  __syncthreads();
  {  // exclusive prefix sum over 4096 cells, 4 per thread
    int v[4], sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i] = s_cnt[4 * t + i];
      sum += v[i];
    }
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int w = 0; w < wave; ++w) base += s_wsum[w];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s_start[4 * t + i] = base;
      base += v[i];
    }
    __syncthreads();
    for (int c = t; c < kCsCells; c += kCsThreads) s_cnt[c] = 0;
    __syncthreads();
  }
  for (int k = t; k < cnt; k += kCsThreads) {
    const float x = in[k * 3 + 0], y = in[k * 3 + 1], z = in[k * 3 + 2];
    const int c = cell_of(x, y, z);
    out.pts[s_start[c] + atomicAdd(&s_cnt[c], 1)] = make_float4(x, y, z, __int_as_float(k));
  }
  for (int k = cnt + t; k < cp; k += kCsThreads)
    out.pts[k] = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), __int_as_float(kCsPad));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int tl = t; tl < cp / kCsTile; tl += kCsThreads) {
    float bl[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float bh[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int i = 0; i < kCsTile; ++i) {
      const float4 p = out.pts[tl * kCsTile + i];
      if (__float_as_int(p.w) != kCsPad) {
        bl[0] = __builtin_fminf(bl[0], p.x); bh[0] = __builtin_fmaxf(bh[0], p.x);
        bl[1] = __builtin_fminf(bl[1], p.y); bh[1] = __builtin_fmaxf(bh[1], p.y);
        bl[2] = __builtin_fminf(bl[2], p.z); bh[2] = __builtin_fmaxf(bh[2], p.z);
      }
    }
    out.tbox[2 * tl + 0] = make_float4(bl[0], bl[1], bl[2], 0.f);
    out.tbox[2 * tl + 1] = make_float4(bh[0], bh[1], bh[2], 0.f);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int bt = t; bt < cp / kCsBatch; bt += kCsThreads) {
    float bl[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float bh[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int i = 0; i < kCsBatch / kCsTile; ++i) {
      const float4 l = out.tbox[2 * (bt * (kCsBatch / kCsTile) + i) + 0];
      const float4 h = out.tbox[2 * (bt * (kCsBatch / kCsTile) + i) + 1];
      bl[0] = __builtin_fminf(bl[0], l.x); bh[0] = __builtin_fmaxf(bh[0], h.x);
      bl[1] = __builtin_fminf(bl[1], l.y); bh[1] = __builtin_fmaxf(bh[1], h.y);
      bl[2] = __builtin_fminf(bl[2], l.z); bh[2] = __builtin_fmaxf(bh[2], h.z);
    }
    out.bbox[2 * bt + 0] = make_float4(bl[0], bl[1], bl[2], 0.f);
    out.bbox[2 * bt + 1] = make_float4(bh[0], bh[1], bh[2], 0.f);
  }
}

template <int Q>
__global__ __launch_bounds__(kCdThreads) void nm_distance_sorted_kernel(
    int n1, int n2, char *__restrict__ scratch, float *__restrict__ dist1,
    float *__restrict__ dist2, int *__restrict__ idx1, int *__restrict__ idx2) {
  const int dir = blockIdx.z;
  const int nq = dir == 0 ? n1 : n2;
  const int nc = dir == 0 ? n2 : n1;
  if ((int)blockIdx.x * (kCdThreads * Q) >= nq) return;
  const int cloud = blockIdx.y;
  const long long per_cloud = cs_side_bytes(n1) + cs_side_bytes(n2);
  char *cbase = scratch + (size_t)cloud * per_cloud;
  const CsSide s1 = cs_carve(cbase, n1), s2 = cs_carve(cbase + cs_side_bytes(n1), n2);
  const CsSide qs = dir == 0 ? s1 : s2, cs = dir == 0 ? s2 : s1;
  float *__restrict__ result = (dir == 0 ? dist1 : dist2) + (size_t)cloud * nq;
  int *__restrict__ result_i = (dir == 0 ? idx1 : idx2) + (size_t)cloud * nq;

  __shared__ __attribute__((aligned(16))) float4 tile[kCsBatch];
  __shared__ __attribute__((aligned(16))) float4 tbx[2 * kCsBatch / kCsTile];

  const int tid = threadIdx.x;
  float qx[Q], qy[Q], qz[Q];
  int qorig[Q];
  unsigned long long best[Q];
  float qlo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
  float qhi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    // a wave's 64*Q queries are consecutive in the sorted order (a compact box)
    int j = blockIdx.x * (kCdThreads * Q) + (tid >> 6) * (64 * Q) + q * 64 + (tid & 63);
    const bool valid = j < nq;
    j = valid ? j : nq - 1;
    const float4 p = qs.pts[j];
    qx[q] = p.x; qy[q] = p.y; qz[q] = p.z;
    qorig[q] = valid ? __float_as_int(p.w) : -1;
    best[q] = ~0ull;
    qlo[0] = __builtin_fminf(qlo[0], p.x); qhi[0] = __builtin_fmaxf(qhi[0], p.x);
    qlo[1] = __builtin_fminf(qlo[1], p.y); qhi[1] = __builtin_fmaxf(qhi[1], p.y);
    qlo[2] = __builtin_fminf(qlo[2], p.z); qhi[2] = __builtin_fmaxf(qhi[2], p.z);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {  // the wave's query box
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      qlo[a] = __builtin_fminf(qlo[a], __shfl_xor(qlo[a], off, 64));
      qhi[a] = __builtin_fmaxf(qhi[a], __shfl_xor(qhi[a], off, 64));
    }
  }
  float wmax = __builtin_inff();  // worst current best distance in this wave

  const int nb = (int)(cs_round_up(nc) / kCsBatch);
  const int nblk = (nq + kCdThreads * Q - 1) / (kCdThreads * Q);
  const int b0 = (int)((long long)blockIdx.x * nb / nblk);  // same relative position in the order
  for (int k = 0; k < 2 * nb; ++k) {
    const int bi = b0 + ((k & 1) ? (k + 1) / 2 : -(k / 2));  // b0, b0+1, b0-1, b0+2, ...
    if (bi < 0 || bi >= nb) continue;                        // block-uniform
    const bool need = !(cs_box_dist(qlo, qhi, cs.bbox[2 * bi], cs.bbox[2 * bi + 1]) > wmax);  // wave-uniform
    if (!__syncthreads_or(need)) continue;
    for (int i = tid; i < kCsBatch; i += kCdThreads) tile[i] = cs.pts[(size_t)bi * kCsBatch + i];
    if (tid < 2 * kCsBatch / kCsTile) tbx[tid] = cs.tbox[(size_t)bi * (2 * kCsBatch / kCsTile) + tid];
    __syncthreads();
    if (need) {
      // lane l tests tile l of the batch (64 tiles); the survivors are visited
      // in order and re-tested against the bests as they tighten
      const float lbl = cs_box_dist(qlo, qhi, tbx[2 * (tid & 63)], tbx[2 * (tid & 63) + 1]);
      // Round 6: the tiles whose box OVERLAPS the wave's query box first -- they hold the queries' neighbourhoods, so the
      // worst best of the wave is tight before the others are tested against it (in index order the first tiles of a
      // batch were evaluated against wmax = inf whether near or far): (64, 16384 x 16384) 0.96 -> 0.81 ms.  (Strictly
      // nearest-first -- a wave minimum per visit -- costs what it saves: 0.94; the decision per 16-lane row: 1.11.)
      const unsigned long long all = __ballot(!(lbl > wmax));
      unsigned long long near = __ballot(lbl == 0.f) & all;
      unsigned long long todo = all & ~near;
      while (near | todo) {
        unsigned long long &from = near ? near : todo;
        const int s = __builtin_ctzll(from);
        from &= from - 1;
        if (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(lbl), s)) > wmax) continue;
        {
          // precise test: does ANY query of the wave still need this tile?  (the
          // same monotone arithmetic on the gaps between the query POINT and
          // the tile's box, against that query's own best)
          const float4 tlo = tbx[2 * s], thi = tbx[2 * s + 1];
          bool mine = false;
#pragma unroll
          for (int q = 0; q < Q; ++q) {
            const float gx = __builtin_fmaxf(__builtin_fmaxf(tlo.x - qx[q], qx[q] - thi.x), 0.f);
            const float gy = __builtin_fmaxf(__builtin_fmaxf(tlo.y - qy[q], qy[q] - thi.y), 0.f);
            const float gz = __builtin_fmaxf(__builtin_fmaxf(tlo.z - qz[q], qz[q] - thi.z), 0.f);
            mine |= !(sqdist3(gx, gy, gz) > __uint_as_float((unsigned)(best[q] >> 32)));
          }
          if (!__any(mine)) continue;
        }
#pragma unroll
        for (int c = 0; c < kCsTile; ++c) {
          const float4 p = tile[s * kCsTile + c];
          const unsigned lo32 = __float_as_uint(p.w);
#pragma unroll
          for (int q = 0; q < Q; ++q) {
            const float d = sqdist3(p.x - qx[q], p.y - qy[q], p.z - qz[q]);
            const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | lo32;
            best[q] = key < best[q] ? key : best[q];
          }
        }
        float m = __uint_as_float((unsigned)(best[0] >> 32));
#pragma unroll
        for (int q = 1; q < Q; ++q) m = __builtin_fmaxf(m, __uint_as_float((unsigned)(best[q] >> 32)));
        wmax = cs_wave_max(m);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    if (qorig[q] >= 0) {
      result[qorig[q]] = __uint_as_float((unsigned)(best[q] >> 32));
      result_i[qorig[q]] = (int)(unsigned)best[q];
    }
  }
}

// Gradient.  One thread per (cloud, point); both directions in one launch.
// grad_xyzA[j] += 2 g (a_j - b_idx), grad_xyzB[idx] -= same (float atomics,
// accumulation order unspecified as in the reference).
__global__ __launch_bounds__(256) void nm_distance_grad_kernel(
    int n1, int n2, const float *__restrict__ xyz1,
    const float *__restrict__ xyz2, const float *__restrict__ graddist1,
    const float *__restrict__ graddist2, const int *__restrict__ idx1,
    const int *__restrict__ idx2, float *__restrict__ gradxyz1,
    float *__restrict__ gradxyz2) {
  const int dir = blockIdx.z;
  const int n = dir == 0 ? n1 : n2;
  const int m = dir == 0 ? n2 : n1;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int cloud = blockIdx.y;
  const float *a = (dir == 0 ? xyz1 : xyz2) + ((size_t)cloud * n + j) * 3;
  const int j2 = (dir == 0 ? idx1 : idx2)[(size_t)cloud * n + j];
  const float *c = (dir == 0 ? xyz2 : xyz1) + ((size_t)cloud * m + j2) * 3;
  const float g = (dir == 0 ? graddist1 : graddist2)[(size_t)cloud * n + j] * 2;
  float *ga = (dir == 0 ? gradxyz1 : gradxyz2) + ((size_t)cloud * n + j) * 3;
  float *gc = (dir == 0 ? gradxyz2 : gradxyz1) + ((size_t)cloud * m + j2) * 3;
  const float gx = g * (a[0] - c[0]);
  const float gy = g * (a[1] - c[1]);
  const float gz = g * (a[2] - c[2]);
  atomicAdd(ga + 0, gx);
  atomicAdd(ga + 1, gy);
  atomicAdd(ga + 2, gz);
  atomicAdd(gc + 0, -gx);
  atomicAdd(gc + 1, -gy);
  atomicAdd(gc + 2, -gz);
}

// The same gradient for training-size clouds (n + m <= 8192 points): one
// workgroup per cloud accumulates both gradient arrays in LDS -- a point's own
// term with a plain add, the scattered term with ds_add_f32 -- and adds the
// result to the (zero-filled) outputs with coalesced, non-atomic writes.  Six
// global float atomics per point (each a full L2 read-modify-write; nearest
// neighbours collide) become six LDS operations: (64, 2048, 2048) 0.19 ->
// 0.03 ms.
constexpr int kCgThreads = 1024;
constexpr int kCgMaxPoints = 8192;  // (n + m) * 12 B of LDS

__global__ __launch_bounds__(kCgThreads) void nm_distance_grad_lds_kernel(
    int n1, int n2, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
    const float *__restrict__ graddist1, const float *__restrict__ graddist2, const int *__restrict__ idx1,
    const int *__restrict__ idx2, float *__restrict__ gradxyz1, float *__restrict__ gradxyz2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *acc1 = reinterpret_cast<float *>(smem);  // n1 * 3
  float *acc2 = acc1 + (size_t)n1 * 3;            // n2 * 3
  const int t = threadIdx.x;
  const int cloud = blockIdx.x;
  xyz1 += (size_t)cloud * n1 * 3;
  xyz2 += (size_t)cloud * n2 * 3;
  for (int i = t; i < (n1 + n2) * 3; i += kCgThreads) acc1[i] = 0.f;
  __syncthreads();
  for (int dir = 0; dir < 2; ++dir) {
    const int n = dir == 0 ? n1 : n2;
    const float *pa = dir == 0 ? xyz1 : xyz2;
    const float *pc = dir == 0 ? xyz2 : xyz1;
    const float *gd = (dir == 0 ? graddist1 : graddist2) + (size_t)cloud * n;
    const int *id = (dir == 0 ? idx1 : idx2) + (size_t)cloud * n;
    float *own = dir == 0 ? acc1 : acc2;
    float *oth = dir == 0 ? acc2 : acc1;
    for (int j = t; j < n; j += kCgThreads) {
      const int j2 = id[j];
      const float g = gd[j] * 2;
      const float gx = g * (pa[j * 3 + 0] - pc[j2 * 3 + 0]);
      const float gy = g * (pa[j * 3 + 1] - pc[j2 * 3 + 1]);
      const float gz = g * (pa[j * 3 + 2] - pc[j2 * 3 + 2]);
      // `own[j]` is also a scatter target of the other direction, which runs
      // in the other pass of this loop (separated by the barrier below)
      own[j * 3 + 0] += gx;
      own[j * 3 + 1] += gy;
      own[j * 3 + 2] += gz;
      atomicAdd(&oth[j2 * 3 + 0], -gx);
      atomicAdd(&oth[j2 * 3 + 1], -gy);
      atomicAdd(&oth[j2 * 3 + 2], -gz);
    }
    __syncthreads();
  }
  float *o1 = gradxyz1 + (size_t)cloud * n1 * 3;
  float *o2 = gradxyz2 + (size_t)cloud * n2 * 3;
  for (int i = t; i < n1 * 3; i += kCgThreads) o1[i] += acc1[i];
  for (int i = t; i < n2 * 3; i += kCgThreads) o2[i] += acc2[i];
}

}  // namespace mvp

using namespace mvp;

extern "C" int mvp_chamfer_forward(int b, int n, int m, const float *xyz1,
                                   const float *xyz2, float *dist1,
                                   float *dist2, int *idx1, int *idx2,
                                   void *stream) {
  if (b < 0 || n < 0 || m < 0) return MVP_EBADSHAPE;
  if (b == 0 || (n == 0 && m == 0)) return MVP_OK;
  if (n == 0 || m == 0) return MVP_EBADSHAPE;  // no nearest neighbour exists
  if (!xyz1 || !xyz2 || !dist1 || !dist2 || !idx1 || !idx2) return MVP_EBADARG;
  if (b > 65535) return MVP_EBADSHAPE;
  const int big = n > m ? n : m;
  // Q queries per lane: 4 when there is enough work to fill 256 CUs anyway.
  const long long work = (long long)b * big;
  if (work >= 4LL * 256 * kCdThreads * 4) {
    dim3 grid((big + kCdThreads * 4 - 1) / (kCdThreads * 4), b, 2);
    hipLaunchKernelGGL(nm_distance_kernel<4>, grid, dim3(kCdThreads), 0,
                       as_stream(stream), n, m, xyz1, xyz2, dist1, dist2, idx1,
                       idx2);
  } else if (work >= 2LL * 256 * kCdThreads * 2) {
    dim3 grid((big + kCdThreads * 2 - 1) / (kCdThreads * 2), b, 2);
    hipLaunchKernelGGL(nm_distance_kernel<2>, grid, dim3(kCdThreads), 0,
                       as_stream(stream), n, m, xyz1, xyz2, dist1, dist2, idx1,
                       idx2);
  } else {
    dim3 grid((big + kCdThreads - 1) / kCdThreads, b, 2);
    hipLaunchKernelGGL(nm_distance_kernel<1>, grid, dim3(kCdThreads), 0,
                       as_stream(stream), n, m, xyz1, xyz2, dist1, dist2, idx1,
                       idx2);
  }
  return check_launch("mvp_chamfer_forward");
}

namespace mvp {
void cs_sort_launch(int b, int n1, int n2, const float *xyz1, const float *xyz2, char *scratch, hipStream_t stream) {
  hipLaunchKernelGGL(chamfer_sort_kernel, dim3(2, b), dim3(kCsThreads), 0, stream, n1, n2, xyz1, xyz2, scratch);
}
}  // namespace mvp

extern "C" long long mvp_chamfer_scratch_bytes(int b, int n, int m) {
  if (b < 0 || n < 0 || m < 0) return -1;
  return (long long)b * (cs_side_bytes(n) + cs_side_bytes(m));
}

extern "C" int mvp_chamfer_forward_sorted(int b, int n, int m, const float *xyz1,
                                          const float *xyz2, float *dist1, float *dist2,
                                          int *idx1, int *idx2, void *scratch,
                                          long long scratch_bytes, void *stream) {
  // small clouds: the exhaustive kernel is already cheaper than sorting
  // (break-even measured at about 4096 x 4096 pairs per cloud)
  if (b <= 0 || n < 2048 || m < 2048 || (long long)n * m < (1ll << 24) || (n > m ? n : m) > (1 << 30))
    return mvp_chamfer_forward(b, n, m, xyz1, xyz2, dist1, dist2, idx1, idx2, stream);
  if (!xyz1 || !xyz2 || !dist1 || !dist2 || !idx1 || !idx2 || !scratch) return MVP_EBADARG;
  if (scratch_bytes < mvp_chamfer_scratch_bytes(b, n, m)) return MVP_EBADARG;
  if ((reinterpret_cast<uintptr_t>(scratch) & 15) != 0) return MVP_EBADARG;
  if (b > 65535) return MVP_EBADSHAPE;
  cs_sort_launch(b, n, m, xyz1, xyz2, reinterpret_cast<char *>(scratch), as_stream(stream));
  // One query per lane: a wave's 64 consecutive sorted queries make the
  // smallest box, so the most tiles are skipped (measured: Q = 1 / 2 / 4 ->
  // 0.88 / 1.02 / 1.35 ms at (64, 16384, 16384); exhaustive 3.95 ms).
  const int big = n > m ? n : m;
  dim3 grid((big + kCdThreads - 1) / kCdThreads, b, 2);
  hipLaunchKernelGGL(nm_distance_sorted_kernel<1>, grid, dim3(kCdThreads), 0, as_stream(stream), n, m,
                     reinterpret_cast<char *>(scratch), dist1, dist2, idx1, idx2);
  return check_launch("mvp_chamfer_forward_sorted");
}

extern "C" int mvp_chamfer_backward(int b, int n, int m, const float *xyz1,
                                    const float *xyz2, float *gradxyz1,
                                    float *gradxyz2, const float *graddist1,
                                    const float *graddist2, const int *idx1,
                                    const int *idx2, void *stream) {
  if (b < 0 || n < 0 || m < 0) return MVP_EBADSHAPE;
  if (b == 0 || (n == 0 && m == 0)) return MVP_OK;
  if (n == 0 || m == 0) return MVP_EBADSHAPE;
  if (!xyz1 || !xyz2 || !gradxyz1 || !gradxyz2 || !graddist1 || !graddist2 ||
      !idx1 || !idx2)
    return MVP_EBADARG;
  if (b > 65535) return MVP_EBADSHAPE;
  if (n + m <= kCgMaxPoints) {
    hipLaunchKernelGGL(nm_distance_grad_lds_kernel, dim3(b), dim3(kCgThreads), (size_t)(n + m) * 12, as_stream(stream),
                       n, m, xyz1, xyz2, graddist1, graddist2, idx1, idx2, gradxyz1, gradxyz2);
    return check_launch("mvp_chamfer_backward");
  }
  const int big = n > m ? n : m;
  dim3 grid((big + 255) / 256, b, 2);
  hipLaunchKernelGGL(nm_distance_grad_kernel, grid, dim3(256), 0,
                     as_stream(stream), n, m, xyz1, xyz2, graddist1, graddist2,
                     idx1, idx2, gradxyz1, gradxyz2);
  return check_launch("mvp_chamfer_backward");
}
