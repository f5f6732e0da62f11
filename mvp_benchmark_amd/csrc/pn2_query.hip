// ball_query / knn / three_nn for gfx950: the brute-force neighbourhood
// queries of the PointNet++ set-abstraction path.
//
// Replaces ball_query_kernel (utils/mm3d_pn2/ops/ball_query/src/
// ball_query_cuda.cu:11-54), knn_kernel (ops/knn/src/knn_cuda.cu:58-95) and
// three_nn_kernel (ops/interpolate/src/three_nn_cuda.cu:11-66).
//
// The reference gives every query one thread that walks global memory
// serially (broadcast, uncoalesced reads).  Here:
//   * knn / three_nn share the Chamfer skeleton: a lane owns a query in
//     registers, candidates stream through a wave-uniform LDS tile;
//   * ball_query uses one WAVE per centre: 64 candidates are tested per step,
//     a ballot + prefix popcount assigns output slots in ascending index
//     order, and the scan stops as soon as nsample hits exist -- the same
//     "first nsample in index order" rule, 64 candidates at a time.
// All three keep the reference's exact selection/tie semantics (see each
// kernel) and the canonical distance chain of the oracle.
#include "common.h"
#include "cs_sort.h"

namespace mvp {

// ------------------------------------------------------------- ball_query
constexpr int kBqThreads = 256;
constexpr int kBqWaves = kBqThreads / kWave;

__global__ __launch_bounds__(kBqThreads) void ball_query_kernel(
    int n, int m, float min_radius, float max_radius, int nsample,
    const float *__restrict__ new_xyz, const float *__restrict__ xyz,
    int *__restrict__ idx) {
  const int wave = threadIdx.x / kWave;
  const int lane = threadIdx.x % kWave;
  const int p = blockIdx.x * kBqWaves + wave;
  if (p >= m) return;  // wave-uniform
  const int cloud = blockIdx.y;
  const float *c = new_xyz + ((size_t)cloud * m + p) * 3;
  const float *pts = xyz + (size_t)cloud * n * 3;
  int *o = idx + ((size_t)cloud * m + p) * nsample;
  const float max_radius2 = max_radius * max_radius;
  const float min_radius2 = min_radius * min_radius;
  const float nx = c[0], ny = c[1], nz = c[2];

  int cnt = 0;
  int first = 0;
  for (int base = 0; base < n && cnt < nsample; base += kWave) {
    const int k = base + lane;
    bool hit = false;
    if (k < n) {
      const float d2 = sqdist3(nx - pts[k * 3 + 0], ny - pts[k * 3 + 1],
                               nz - pts[k * 3 + 2]);
      hit = d2 == 0 || (d2 >= min_radius2 && d2 < max_radius2);
    }
    const unsigned long long mask = __ballot(hit);
    if (mask == 0) continue;
    if (cnt == 0) first = base + __builtin_ctzll(mask);
    const int pos = cnt + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
    if (hit && pos < nsample) o[pos] = k;
    cnt += __builtin_popcountll(mask);
  }
  // Slots beyond the hits keep the first hit (ball_query_cuda.cu:44-48); a
  // centre with no hit keeps the caller's zeros (ball_query.py:35).
  const int filled = cnt < nsample ? cnt : nsample;
  const int fill = cnt > 0 ? first : 0;
  for (int l = filled + lane; l < nsample; l += kWave) o[l] = fill;
}

// -------------------------------------------------------------------- knn
constexpr int kQTile = 1024;  // candidates per LDS tile (a multiple of 32)

// Max-heap in LDS, one column per thread: slot s of thread t at [s*T + t].
template <int T>
__device__ __forceinline__ void knn_reheap(float *hd, int *hi, int t, int k) {
  int root = 0;
  int child = 1;
  while (child < k) {
    if (child + 1 < k && hd[(child + 1) * T + t] > hd[child * T + t]) child++;
    if (hd[root * T + t] > hd[child * T + t]) return;
    const float tf = hd[root * T + t];
    hd[root * T + t] = hd[child * T + t];
    hd[child * T + t] = tf;
    const int ti = hi[root * T + t];
    hi[root * T + t] = hi[child * T + t];
    hi[child * T + t] = ti;
    root = child;
    child = root * 2 + 1;
  }
}

// knn_kernel, knn_cuda.cu:58-95: admission by strict `<` against the heap
// root (:83), reheap swapping on equality (:34), final heap_sort (:44-53).
template <int T>
__global__ __launch_bounds__(T) void knn_kernel(
    int n, int m, int nsample, const float *__restrict__ xyz,
    const float *__restrict__ new_xyz, int *__restrict__ idx,
    float *__restrict__ dist2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *tile = reinterpret_cast<float *>(smem);              // kQTile*3
  float *hd = tile + kQTile * 3;                              // nsample*T
  int *hi = reinterpret_cast<int *>(hd + (size_t)nsample * T);  // nsample*T

  const int t = threadIdx.x;
  const int cloud = blockIdx.y;
  const int p = blockIdx.x * T + t;
  const bool active = p < m;
  const int pc = active ? p : m - 1;
  const float *c = new_xyz + ((size_t)cloud * m + pc) * 3;
  const float *pts = xyz + (size_t)cloud * n * 3;
  const float nx = c[0], ny = c[1], nz = c[2];

  for (int i = 0; i < nsample; ++i) {
    hd[i * T + t] = 1e10f;
    hi[i * T + t] = 0;
  }
  float rootd = 1e10f;

  // Candidates are screened 32 at a time against the heap root the sub-tile
  // started with (branch-free, three wave-uniform ds_read_b128 per four
  // candidates); the survivors -- a superset of what the reference admits,
  // because the root only falls -- are then replayed in index order against
  // the CURRENT root, so the sequence of heap operations is the reference's.
  // A wave therefore runs the divergent heap update max-over-lanes(survivors)
  // times per sub-tile instead of once per candidate that any lane admits.
  for (int k2 = 0; k2 < n; k2 += kQTile) {
    const int cnt = min(kQTile, n - k2);
    const int padded = (cnt + 31) / 32 * 32;
    __syncthreads();
    for (int q = t; q < padded * 3; q += T)
      tile[q] = q < cnt * 3 ? pts[(size_t)k2 * 3 + q] : __builtin_inff();  // d2 = +inf: never admitted
    __syncthreads();
    const float4 *__restrict__ t4 = reinterpret_cast<const float4 *>(tile);
    for (int sub = 0; sub < padded; sub += 32) {
      unsigned mask = 0u;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const float4 a = t4[(sub / 4 + g) * 3 + 0];
        const float4 b = t4[(sub / 4 + g) * 3 + 1];
        const float4 c = t4[(sub / 4 + g) * 3 + 2];
        const float d0 = sqdist3(nx - a.x, ny - a.y, nz - a.z);
        const float d1 = sqdist3(nx - a.w, ny - b.x, nz - b.y);
        const float d2 = sqdist3(nx - b.z, ny - b.w, nz - c.x);
        const float d3 = sqdist3(nx - c.y, ny - c.z, nz - c.w);
        mask |= (d0 < rootd ? 1u : 0u) << (4 * g + 0);
        mask |= (d1 < rootd ? 1u : 0u) << (4 * g + 1);
        mask |= (d2 < rootd ? 1u : 0u) << (4 * g + 2);
        mask |= (d3 < rootd ? 1u : 0u) << (4 * g + 3);
      }
      while (__any(mask != 0u)) {
        if (mask != 0u) {
          const int i = sub + __builtin_ctz(mask);
          mask &= mask - 1u;
          const float d2 = sqdist3(nx - tile[i * 3 + 0], ny - tile[i * 3 + 1], nz - tile[i * 3 + 2]);
          if (d2 < rootd) {
            hd[t] = d2;
            hi[t] = k2 + i;
            knn_reheap<T>(hd, hi, t, nsample);
            rootd = hd[t];
          }
        }
      }
    }
  }
  // heap_sort
  for (int i = nsample - 1; i > 0; i--) {
    const float tf = hd[t];
    hd[t] = hd[i * T + t];
    hd[i * T + t] = tf;
    const int ti = hi[t];
    hi[t] = hi[i * T + t];
    hi[i * T + t] = ti;
    knn_reheap<T>(hd, hi, t, i);
  }
  if (active) {
    int *o = idx + ((size_t)cloud * m + p) * nsample;
    float *od = dist2 + ((size_t)cloud * m + p) * nsample;
    for (int i = 0; i < nsample; ++i) {
      o[i] = hi[i * T + t];
      od[i] = hd[i * T + t];
    }
  }
}

// One WAVE per listed query: the reference's sequence of heap operations (admission in index order by strict
// `<` against the root, reheap, final heap sort: knn_cuda.cu:26-53,80-90) with the 64 lanes screening 64
// consecutive candidates at a time against the root the step started with -- the survivors, a superset of
// what the reference admits there, are replayed in index order against the current root.  The fix-up pass of
// the sorted variant below: a handful of queries per million on random clouds, every query on a lattice.
constexpr int kFixWaves = 4;
__global__ __launch_bounds__(kFixWaves * 64) void knn_fix_kernel(
    int n, int m, int nsample, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    int *__restrict__ idx, float *__restrict__ dist2, const int *__restrict__ fix_cnt, const int *__restrict__ fix_list) {
  __shared__ float s_hd[kFixWaves][32];
  __shared__ int s_hi[kFixWaves][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cloud = blockIdx.y;
  const int cnt = min(fix_cnt[cloud], m);
  float *hd = s_hd[wave];
  int *hi = s_hi[wave];
  const float *pts = xyz + (size_t)cloud * n * 3;
  for (int e = blockIdx.x * kFixWaves + wave; e < cnt; e += gridDim.x * kFixWaves) {
    const int p = fix_list[(size_t)cloud * m + e];
    const float *c = new_xyz + ((size_t)cloud * m + p) * 3;
    const float nx = c[0], ny = c[1], nz = c[2];
    if (lane < nsample) {
      hd[lane] = 1e10f;
      hi[lane] = 0;
    }
    float rootd = 1e10f;
    for (int base = 0; base < n; base += 64) {
      const int i = base + lane;
      float d = __builtin_inff();
      if (i < n) d = sqdist3(nx - pts[(size_t)i * 3 + 0], ny - pts[(size_t)i * 3 + 1], nz - pts[(size_t)i * 3 + 2]);
      unsigned long long mask = __ballot(d < rootd);
      while (mask) {
        const int l = (int)__builtin_ctzll(mask);
        mask &= mask - 1ull;
        const float dl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), l));
        if (dl < rootd) {   // (wave-uniform: every lane runs the same heap update on the wave's LDS column)
          hd[0] = dl;
          hi[0] = base + l;
          knn_reheap<1>(hd, hi, 0, nsample);
          rootd = hd[0];
        }
      }
    }
    for (int i = nsample - 1; i > 0; i--) {   // heap_sort
      const float tf = hd[0];
      hd[0] = hd[i];
      hd[i] = tf;
      const int ti = hi[0];
      hi[0] = hi[i];
      hi[i] = ti;
      knn_reheap<1>(hd, hi, 0, i);
    }
    if (lane < nsample) {
      idx[((size_t)cloud * m + p) * nsample + lane] = hi[lane];
      dist2[((size_t)cloud * m + p) * nsample + lane] = hd[lane];
    }
  }
}

// Pruned exact variant for large clouds (n >= 4096, k <= 32): the same result with most of the m*n pairs
// never evaluated.  Both point sets are Morton-sorted with boxes per 16 and per 1024 points (cs_sort.h,
// chamfer_sort_kernel); a lane owns a query -- a wave's 64 consecutive sorted queries sit in a small box --
// and keeps a max-heap of its k+1 best candidates in an LDS column; batches are visited outwards from the
// query block's own position in the order, a batch / tile is skipped when its box is farther from the
// wave's query box than the worst heap root of the wave, and a tile is evaluated only if some query's own
// root still reaches its box (strict `>` on monotone box distances, as in the sorted Chamfer kernel).
// Exactness.  The reference admits candidates in INDEX order by strict `<` against the heap root and ends
// with a heap sort (knn_cuda.cu:80-90): when the k+1 smallest distances of a query are pairwise different,
// the k nearest are a unique set and the heap sort returns them in ascending order whatever the order of
// admission was -- that is what this kernel computes in its own visiting order.  When two of them are EQUAL
// (lattices, duplicated points) both which candidates stay and where they end up depend on the reference's
// sequence of heap operations: such queries are flagged and recomputed by knn_kernel, which replays that
// sequence (knn_fix_kernel, one wave per listed query).
// (The k+1 candidates of a query live in REGISTERS as an ascending list -- insertion is a branch-free
// compare/select chain over KL compile-time slots, KL >= nsample + 1; keeping more than needed only loosens
// the pruning bound.  The exhaustive kernel's LDS heaps, which the reference's tie behaviour needs, cost a
// dependent LDS round trip per heap level and admission: most of that kernel's time.)
template <int T, int KL>
__global__ __launch_bounds__(T) void knn_sorted_kernel(
    int n, int m, int nsample, char *__restrict__ scratch, int *__restrict__ idx, float *__restrict__ dist2,
    int *__restrict__ fix_cnt, int *__restrict__ fix_list) {
  __shared__ __attribute__((aligned(16))) float4 tile[kCsBatch];                 // one batch of candidates
  __shared__ __attribute__((aligned(16))) float4 tbx[2 * kCsBatch / kCsTile];    // its tiles' boxes
  const int K1 = nsample + 1;

  const int t = threadIdx.x;
  const int cloud = blockIdx.y;
  const long long per_cloud = cs_side_bytes(m) + cs_side_bytes(n);
  char *cbase = scratch + (size_t)cloud * per_cloud;
  const CsSide qs = cs_carve(cbase, m), cs = cs_carve(cbase + cs_side_bytes(m), n);
  int j = blockIdx.x * T + t;   // consecutive in the sorted order
  const bool valid = j < m;
  j = valid ? j : m - 1;
  const float4 q = qs.pts[j];
  const int qorig = valid ? __float_as_int(q.w) : -1;
  const float qx = q.x, qy = q.y, qz = q.z;
  float qlo[3] = {qx, qy, qz}, qhi[3] = {qx, qy, qz};
#pragma unroll
  for (int a = 0; a < 3; ++a) {  // the wave's query box
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      qlo[a] = __builtin_fminf(qlo[a], __shfl_xor(qlo[a], off, 64));
      qhi[a] = __builtin_fmaxf(qhi[a], __shfl_xor(qhi[a], off, 64));
    }
  }
  float dd[KL];
  int ii[KL];
#pragma unroll
  for (int i = 0; i < KL; ++i) {
    dd[i] = 1e10f;
    ii[i] = 0;
  }
  float wmax = 1e10f;   // worst list tail of this wave

  const int nb = (int)(cs_round_up(n) / kCsBatch);
  const int nblk = (m + T - 1) / T;
  const int b0 = (int)((long long)blockIdx.x * nb / nblk);  // same relative position in the order
  // the tile at the wave's own relative position in the candidates' order
  const int own_tile = (int)(((long long)blockIdx.x * T + (t & ~63) + 32) * (long long)(cs_round_up(n) / kCsTile) / max(1, m));
  for (int k = 0; k < 2 * nb; ++k) {
    const int bi = b0 + ((k & 1) ? (k + 1) / 2 : -(k / 2));  // b0, b0+1, b0-1, b0+2, ...
    if (bi < 0 || bi >= nb) continue;                        // block-uniform
    const bool need = !(cs_box_dist(qlo, qhi, cs.bbox[2 * bi], cs.bbox[2 * bi + 1]) > wmax);  // wave-uniform
    if (!__syncthreads_or(need)) continue;
    for (int i = t; i < kCsBatch; i += T) tile[i] = cs.pts[(size_t)bi * kCsBatch + i];
    for (int i = t; i < 2 * kCsBatch / kCsTile; i += T) tbx[i] = cs.tbox[(size_t)bi * (2 * kCsBatch / kCsTile) + i];
    __syncthreads();
    if (need) {
      const float lbl = cs_box_dist(qlo, qhi, tbx[2 * (t & 63)], tbx[2 * (t & 63) + 1]);   // lane l: tile l of the batch
      unsigned long long todo = __ballot(!(lbl > wmax));
      // tiles are taken outwards from the wave's own relative position in the order (alternately the next one
      // above and the next one below it): the lists tighten on the nearest tiles first
      const int own = min(63, max(0, own_tile - bi * (kCsBatch / kCsTile)));
      bool up = true;
      while (todo) {
        const unsigned long long hi_part = todo & (~0ull << own), lo_part = todo & ~(~0ull << own);
        int s;
        if ((up && hi_part) || !lo_part) s = (int)__builtin_ctzll(hi_part);
        else s = 63 - (int)__builtin_clzll(lo_part);
        up = !up;
        todo &= ~(1ull << s);
        if (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(lbl), s)) > wmax) continue;
        const float4 tlo = tbx[2 * s], thi = tbx[2 * s + 1];
        const float gx = __builtin_fmaxf(__builtin_fmaxf(tlo.x - qx, qx - thi.x), 0.f);
        const float gy = __builtin_fmaxf(__builtin_fmaxf(tlo.y - qy, qy - thi.y), 0.f);
        const float gz = __builtin_fmaxf(__builtin_fmaxf(tlo.z - qz, qz - thi.z), 0.f);
        if (!__any(!(sqdist3(gx, gy, gz) > dd[KL - 1]))) continue;   // no query's list reaches this tile
        // (every candidate that ANY lane admits runs the insertion for the whole wave; letting each lane take its
        // own next survivor -- max-over-lanes(survivors) insertions per tile -- was slower: 0.93 -> 2.09 ms, the
        // per-lane LDS gather and the loop around it cost more than the insertions they save)
        bool changed = false;
#pragma unroll 4
        for (int c = 0; c < kCsTile; ++c) {
          const float4 p = tile[s * kCsTile + c];
          const float d = sqdist3(qx - p.x, qy - p.y, qz - p.z);
          if (__any(d < dd[KL - 1])) {
            const int id = __float_as_int(p.w);
#pragma unroll
            for (int i = KL - 1; i >= 1; --i) {
              const bool sh = d < dd[i - 1];
              const bool here = !sh && d < dd[i];
              dd[i] = sh ? dd[i - 1] : (here ? d : dd[i]);
              ii[i] = sh ? ii[i - 1] : (here ? id : ii[i]);
            }
            const bool here0 = d < dd[0];
            dd[0] = here0 ? d : dd[0];
            ii[0] = here0 ? id : ii[0];
            changed = true;
          }
        }
        if (changed) wmax = cs_wave_max(dd[KL - 1]);   // (wave-uniform)
      }
    }
    __syncthreads();
  }
  // are the k+1 smallest distances pairwise different?  (the list is ascending)
  if (valid) {
    bool tie = false;
#pragma unroll
    for (int i = 0; i + 1 < KL; ++i) tie |= i + 1 < K1 && dd[i] == dd[i + 1];
    if (tie) fix_list[(size_t)cloud * m + atomicAdd(&fix_cnt[cloud], 1)] = qorig;
    int *o = idx + ((size_t)cloud * m + qorig) * nsample;
    float *od = dist2 + ((size_t)cloud * m + qorig) * nsample;
#pragma unroll
    for (int i = 0; i < KL; ++i) {
      if (i < nsample) {
        o[i] = ii[i];
        od[i] = dd[i];
      }
    }
  }
}

// --------------------------------------------------------------- three_nn
constexpr int kNnThreads = 256;

// three_nn_kernel, three_nn_cuda.cu:11-66.  The reference keeps the three
// bests in double initialised to 1e40 and stores them as float; comparing a
// float d against such a double is identical to a float compare against +inf
// (1e40 rounds to +inf on store), so float registers with +inf are exact.
__global__ __launch_bounds__(kNnThreads) void three_nn_kernel(
    int n, int m, const float *__restrict__ unknown,
    const float *__restrict__ known, float *__restrict__ dist2,
    int *__restrict__ idx) {
  __shared__ __attribute__((aligned(16))) float tile[kQTile * 3];
  const int t = threadIdx.x;
  const int cloud = blockIdx.y;
  const int p = blockIdx.x * kNnThreads + t;
  const bool active = p < n;
  const int pc = active ? p : n - 1;
  const float *u = unknown + ((size_t)cloud * n + pc) * 3;
  const float *kn = known + (size_t)cloud * m * 3;
  const float ux = u[0], uy = u[1], uz = u[2];
  float best1 = __builtin_inff(), best2 = __builtin_inff(),
        best3 = __builtin_inff();
  int besti1 = 0, besti2 = 0, besti3 = 0;
  for (int k2 = 0; k2 < m; k2 += kQTile) {
    const int cnt = min(kQTile, m - k2);
    __syncthreads();
    for (int q = t; q < cnt * 3; q += kNnThreads) tile[q] = kn[(size_t)k2 * 3 + q];
    __syncthreads();
    for (int i = 0; i < cnt; ++i) {
      const float d = sqdist3(ux - tile[i * 3 + 0], uy - tile[i * 3 + 1],
                              uz - tile[i * 3 + 2]);
      const int k = k2 + i;
      if (d < best3) {  // common case (no change) costs one compare
        if (d < best1) {
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = d; besti1 = k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2;
          best2 = d; besti2 = k;
        } else {
          best3 = d; besti3 = k;
        }
      }
    }
  }
  if (active) {
    const size_t o = ((size_t)cloud * n + p) * 3;
    dist2[o + 0] = best1;
    dist2[o + 1] = best2;
    dist2[o + 2] = best3;
    idx[o + 0] = besti1;
    idx[o + 1] = besti2;
    idx[o + 2] = besti3;
  }
}

// Feature-space neighbours of the models: the k largest of
//   val[i][j] = fl(fl(-sq[j] - (-2 * dot[i][j])) - sq[i])
// per row i of a (N x N) Gram matrix `dot` = x^T x that the caller computed
// with a library GEMM -- exactly the expression model_utils.knn builds
// (completion/model_utils.py:242-247: `-xx - inner - xx.transpose(2, 1)`), so
// the selected indices equal torch.topk's wherever the values are distinct.
// Replaces three elementwise passes over the (B,N,N) matrix plus a radix
// top-k by one pass: a block owns 128 rows, tiles of 128 x 32 values are
// loaded with coalesced row reads and handed to the row's lane through LDS
// (row stride 33: conflict-free), and each lane runs the knn kernel's
// screened heap on key = -val.
constexpr int kFkRows = 128;
constexpr int kFkCols = 32;

__global__ __launch_bounds__(kFkRows) void topk_gram_kernel(
    int n, int k, const float *__restrict__ dot, const float *__restrict__ sq,
    int *__restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *tile = reinterpret_cast<float *>(smem);                 // kFkRows * (kFkCols + 1)
  float *sqc = tile + kFkRows * (kFkCols + 1);                    // kFkCols
  float *hd = sqc + kFkCols;                                      // k * kFkRows
  int *hi = reinterpret_cast<int *>(hd + (size_t)k * kFkRows);   // k * kFkRows
  const int t = threadIdx.x;
  const int cloud = blockIdx.y;
  const int row0 = blockIdx.x * kFkRows;
  const int row = row0 + t;
  const bool active = row < n;
  dot += (size_t)cloud * n * n;
  sq += (size_t)cloud * n;
  const float sqi = active ? sq[row] : 0.f;
  for (int i = 0; i < k; ++i) {
    hd[i * kFkRows + t] = __builtin_inff();
    hi[i * kFkRows + t] = 0;
  }
  float rootd = __builtin_inff();
  for (int c0 = 0; c0 < n; c0 += kFkCols) {
    __syncthreads();
    // 128 x 32 tile: thread (r = e / 32, c = e % 32) -> 32 consecutive floats per row
    for (int e = t; e < kFkRows * kFkCols; e += kFkRows) {
      const int r = e / kFkCols, c = e % kFkCols;
      const bool ok = row0 + r < n && c0 + c < n;
      tile[r * (kFkCols + 1) + c] = ok ? dot[(size_t)(row0 + r) * n + c0 + c] : 0.f;
    }
    if (t < kFkCols) sqc[t] = c0 + t < n ? sq[c0 + t] : __builtin_inff();   // key = +inf: never admitted
    __syncthreads();
    unsigned mask = 0u;
#pragma unroll
    for (int c = 0; c < kFkCols; ++c) {
      const float inner = -2.f * tile[t * (kFkCols + 1) + c];
      const float key = -((-sqc[c] - inner) - sqi);
      mask |= (key < rootd ? 1u : 0u) << c;
    }
    while (__any(mask != 0u)) {
      if (mask != 0u) {
        const int c = __builtin_ctz(mask);
        mask &= mask - 1u;
        const float inner = -2.f * tile[t * (kFkCols + 1) + c];
        const float key = -((-sqc[c] - inner) - sqi);
        if (key < rootd) {
          hd[t] = key;
          hi[t] = c0 + c;
          knn_reheap<kFkRows>(hd, hi, t, k);
          rootd = hd[t];
        }
      }
    }
  }
  for (int i = k - 1; i > 0; i--) {  // heap sort: ascending key = descending value
    const float tf = hd[t];
    hd[t] = hd[i * kFkRows + t];
    hd[i * kFkRows + t] = tf;
    const int ti = hi[t];
    hi[t] = hi[i * kFkRows + t];
    hi[i * kFkRows + t] = ti;
    knn_reheap<kFkRows>(hd, hi, t, i);
  }
  if (active) {
    int *o = idx + ((size_t)cloud * n + row) * k;
    for (int i = 0; i < k; ++i) o[i] = hi[i * kFkRows + t];
  }
}


// The same selection with one WAVE per row (k <= 64): the row is streamed
// straight from the Gram matrix (no LDS transpose, no workgroup barriers in the
// scan; VEC = 4: four 16-byte loads per lane = 4 KB per wave in flight), the
// running top-k lives SORTED in lanes 0..k-1 of two registers, candidates are
// screened 64 at a time against the k-th key with one ballot, and a survivor
// is inserted with a one-lane shift (DPP wave_shr) of the entries behind it.
// After the first few hundred columns a row admits a candidate every ~35
// columns (k ln(n/k) insertions in total), so the kernel is a stream of the
// 4 n^2 bytes.  The list is ordered by (key, column): equal keys keep the lower
// column first whatever the visiting order.
constexpr int kTwWaves = 4;   // waves (= rows in flight) per workgroup
constexpr int kTwRows = 8;    // consecutive rows per wave

template <int VEC>
__global__ __launch_bounds__(kTwWaves * kWave) void topk_gram_wave_kernel(
    int n, int k, const float *__restrict__ dot, const float *__restrict__ sq, int *__restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *s_sq = reinterpret_cast<float *>(smem);  // n rounded up to 4
  const int t = threadIdx.x, lane = t & (kWave - 1), wave = t >> 6;
  const int cloud = blockIdx.y;
  dot += (size_t)cloud * n * n;
  sq += (size_t)cloud * n;
  for (int i = t; i < n; i += kTwWaves * kWave) s_sq[i] = sq[i];
  __syncthreads();
  const int row0 = (blockIdx.x * kTwWaves + wave) * kTwRows;
  for (int row = row0; row < min(row0 + kTwRows, n); ++row) {
    const float *__restrict__ drow = dot + (size_t)row * n;
    const float sqi = s_sq[row];
    float lv = __builtin_inff();  // key = -value, ascending in lanes 0..k-1; lanes >= k never shift
    int li = 0;  // slots never filled (fewer than k finite keys: NaN input) point at column 0
    float tau = __builtin_inff();  // exclusive bound: a candidate must be strictly smaller
    // Until k entries are in, the list's k-th key is +inf and everything would
    // pass the screen.  The first block of columns gives a bound for free: the
    // largest of the 64 per-lane minima has >= 64 >= k keys at or below it.
    float seed = __builtin_inff();
    auto insert = [&](float kv, int col) {
      const bool gt = lane < k && (lv > kv || (lv == kv && li > col));  // entries that move one lane up
      // wave_shr:1 -- lane l reads lane l-1 (lane 0: 0), one DPP move each
      const float plv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(lv), 0x138, 0xF, 0xF, true));
      const int pli = __builtin_amdgcn_update_dpp(0, li, 0x138, 0xF, 0xF, true);
      const bool pgt = __builtin_amdgcn_update_dpp(0, gt ? 1 : 0, 0x138, 0xF, 0xF, true) != 0;
      lv = gt ? (pgt ? plv : kv) : lv;
      li = gt ? (pgt ? pli : col) : li;
      tau = __builtin_fminf(seed, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lv), k - 1)));
    };
    // seed = the k-th smallest of the 64 per-lane minima of the first block (k
    // keys at or below it): radix select on the order-preserving bit pattern,
    // one ballot per bit, all bookkeeping scalar.
    auto seed_from = [&](float lane_min) {
      const unsigned fb = __float_as_uint(lane_min);
      const unsigned ord = (fb & 0x80000000u) ? ~fb : (fb | 0x80000000u);
      unsigned long long cand = ~0ull;
      unsigned prefix = 0u;
      int kk = k;
#pragma unroll
      for (int bit = 31; bit >= 0; --bit) {
        const unsigned long long zeros = cand & ~__ballot((ord >> bit) & 1u);
        const int cz = __builtin_popcountll(zeros);
        if (kk <= cz) {
          cand = zeros;
        } else {
          kk -= cz;
          cand &= ~zeros;
          prefix |= 1u << bit;
        }
      }
      const unsigned pb = (prefix & 0x80000000u) ? (prefix & 0x7FFFFFFFu) : ~prefix;
      const float kth = __uint_as_float(pb);
      // next float above kth (finite): the bound is exclusive
      const int mb = __float_as_int(kth == 0.f ? 0.f : kth);
      seed = kth < __builtin_inff() ? __int_as_float(mb >= 0 ? mb + 1 : mb - 1) : kth;
      tau = seed;
    };
    auto screen = [&](float key, int colbase, int colstride) {  // lane's candidate: column colbase + colstride * lane
      unsigned long long m = __ballot(key < tau);
      while (m) {
        const int l = __builtin_ctzll(m);
        m &= m - 1ull;
        const float kv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(key), l));
        if (kv < tau) insert(kv, colbase + colstride * l);  // wave-uniform
      }
    };
    if constexpr (VEC == 4) {  // n % 4 == 0: rows are 16-byte aligned
      for (int c0 = 0; c0 < n; c0 += 4 * 4 * kWave) {
        float4 d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = c0 + (u * kWave + lane) * 4;
          d[u] = c < n ? *reinterpret_cast<const float4 *>(drow + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float key[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = c0 + (u * kWave + lane) * 4;
          const bool ok = c < n;
          const float4 q = ok ? *reinterpret_cast<const float4 *>(s_sq + c) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float inf = __builtin_inff();
          key[u][0] = ok ? -((-q.x - (-2.f * d[u].x)) - sqi) : inf;
          key[u][1] = ok ? -((-q.y - (-2.f * d[u].y)) - sqi) : inf;
          key[u][2] = ok ? -((-q.z - (-2.f * d[u].z)) - sqi) : inf;
          key[u][3] = ok ? -((-q.w - (-2.f * d[u].w)) - sqi) : inf;
        }
        if (c0 == 0) {
          float lm = key[0][0];
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) lm = __builtin_fminf(lm, key[u][v]);
          seed_from(lm);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int v = 0; v < 4; ++v) screen(key[u][v], c0 + u * kWave * 4 + v, 4);
      }
    } else {
      for (int c0 = 0; c0 < n; c0 += 4 * kWave) {
        float key[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = c0 + u * kWave + lane;
          const float d = c < n ? drow[c] : 0.f;
          key[u] = c < n ? -((-s_sq[c] - (-2.f * d)) - sqi) : __builtin_inff();
        }
        if (c0 == 0) seed_from(__builtin_fminf(__builtin_fminf(key[0], key[1]), __builtin_fminf(key[2], key[3])));
#pragma unroll
        for (int u = 0; u < 4; ++u) screen(key[u], c0 + u * kWave, 1);
      }
    }
    if (lane < k) idx[((size_t)cloud * n + row) * k + lane] = li;
  }
}

}  // namespace mvp

using namespace mvp;

extern "C" int mvp_ball_query(int b, int n, int m, float min_radius,
                              float max_radius, int nsample,
                              const float *new_xyz, const float *xyz, int *idx,
                              void *stream) {
  if (b < 0 || n < 0 || m < 0 || nsample < 0) return MVP_EBADSHAPE;
  if (b == 0 || m == 0 || nsample == 0) return MVP_OK;
  if (!new_xyz || !idx || (n > 0 && !xyz)) return MVP_EBADARG;
  if (b > 65535) return MVP_EBADSHAPE;
  dim3 grid((m + kBqWaves - 1) / kBqWaves, b);
  hipLaunchKernelGGL(ball_query_kernel, grid, dim3(kBqThreads), 0,
                     as_stream(stream), n, m, min_radius, max_radius, nsample,
                     new_xyz, xyz, idx);
  return check_launch("mvp_ball_query");
}

extern "C" int mvp_knn(int b, int n, int m, int nsample, const float *xyz,
                       const float *new_xyz, int *idx, float *dist2,
                       void *stream) {
  if (b < 0 || n < 0 || m < 0 || nsample < 1 || nsample > 100)
    return MVP_EBADSHAPE;
  if (b == 0 || m == 0) return MVP_OK;
  if (!new_xyz || !idx || !dist2 || (n > 0 && !xyz)) return MVP_EBADARG;
  if (b > 65535) return MVP_EBADSHAPE;
  // Heap columns live in LDS; block size shrinks with k so that
  // tile (12 KiB) + heaps (nsample*T*8 B) stays under 64 KiB.
  if (nsample <= 16) {
    constexpr int T = 256;
    const size_t lds = kQTile * 3 * 4 + (size_t)nsample * T * 8;
    dim3 grid((m + T - 1) / T, b);
    hipLaunchKernelGGL(knn_kernel<T>, grid, dim3(T), lds, as_stream(stream), n,
                       m, nsample, xyz, new_xyz, idx, dist2);
  } else if (nsample <= 32) {
    constexpr int T = 128;
    const size_t lds = kQTile * 3 * 4 + (size_t)nsample * T * 8;
    dim3 grid((m + T - 1) / T, b);
    hipLaunchKernelGGL(knn_kernel<T>, grid, dim3(T), lds, as_stream(stream), n,
                       m, nsample, xyz, new_xyz, idx, dist2);
  } else {
    constexpr int T = 64;
    const size_t lds = kQTile * 3 * 4 + (size_t)nsample * T * 8;
    dim3 grid((m + T - 1) / T, b);
    hipLaunchKernelGGL(knn_kernel<T>, grid, dim3(T), lds, as_stream(stream), n,
                       m, nsample, xyz, new_xyz, idx, dist2);
  }
  return check_launch("mvp_knn");
}

// scratch of mvp_knn_sorted: the sorted sides of every cloud, then b counters (queries to recompute) padded to
// 16 bytes, then b x m query ids
static long long knn_cnt_bytes(int b) { return ((long long)b * 4 + 15) / 16 * 16; }
extern "C" long long mvp_knn_scratch_bytes(int b, int n, int m) {
  if (b < 0 || n < 0 || m < 0) return -1;
  return (long long)b * (cs_side_bytes(m) + cs_side_bytes(n)) + knn_cnt_bytes(b) + ((long long)b * m * 4 + 15) / 16 * 16;
}

extern "C" int mvp_knn_sorted(int b, int n, int m, int nsample, const float *xyz, const float *new_xyz, int *idx,
                              float *dist2, void *scratch, long long scratch_bytes, void *stream) {
  // small clouds / many neighbours: the exhaustive kernel (the heaps of k + 1 <= 33 entries per query fit the
  // workgroup's LDS)
  // (round 4, at the networks' sizes: 3072 x 3072, k = 20: 0.67 against 0.93 ms, 2048 x 2048, k = 16: 0.28 against 0.32;
  // 3072 -> 1536 and 1536 x 1536 stay with the exhaustive kernel: 0.40 / 0.32 against 0.44 / 0.56)
  const bool large = (n >= 4096 && m >= 1024) || (n >= 2048 && m >= 2048);
  if (b <= 0 || !large || nsample < 1 || nsample > 32 || (n > m ? n : m) > (1 << 30))
    return mvp_knn(b, n, m, nsample, xyz, new_xyz, idx, dist2, stream);
  if (!xyz || !new_xyz || !idx || !dist2 || !scratch) return MVP_EBADARG;
  if (scratch_bytes < mvp_knn_scratch_bytes(b, n, m)) return MVP_EBADARG;
  if ((reinterpret_cast<uintptr_t>(scratch) & 15) != 0) return MVP_EBADARG;
  if (b > 65535) return MVP_EBADSHAPE;
  char *sc = reinterpret_cast<char *>(scratch);
  int *fix_cnt = reinterpret_cast<int *>(sc + (size_t)b * (cs_side_bytes(m) + cs_side_bytes(n)));
  int *fix_list = reinterpret_cast<int *>(reinterpret_cast<char *>(fix_cnt) + knn_cnt_bytes(b));
  if (hipMemsetAsync(fix_cnt, 0, (size_t)knn_cnt_bytes(b), as_stream(stream)) != hipSuccess) return check_launch("mvp_knn_sorted");
  cs_sort_launch(b, m, n, new_xyz, xyz, sc, as_stream(stream));   // side 0: the queries, side 1: the candidates
  constexpr int T = 256;
  const dim3 grid((m + T - 1) / T, b);
  if (nsample <= 8)
    hipLaunchKernelGGL((knn_sorted_kernel<T, 9>), grid, dim3(T), 0, as_stream(stream), n, m, nsample, sc, idx, dist2, fix_cnt, fix_list);
  else if (nsample <= 16)
    hipLaunchKernelGGL((knn_sorted_kernel<T, 17>), grid, dim3(T), 0, as_stream(stream), n, m, nsample, sc, idx, dist2, fix_cnt, fix_list);
  else
    hipLaunchKernelGGL((knn_sorted_kernel<T, 33>), grid, dim3(T), 0, as_stream(stream), n, m, nsample, sc, idx, dist2, fix_cnt, fix_list);
  // queries with equal distances among their k + 1 nearest: the reference's own sequence of heap operations
  hipLaunchKernelGGL(knn_fix_kernel, dim3(16, b), dim3(kFixWaves * 64), 0, as_stream(stream), n, m, nsample, xyz, new_xyz, idx,
                     dist2, fix_cnt, fix_list);
  return check_launch("mvp_knn_sorted");
}

extern "C" int mvp_topk_gram(int b, int n, int k, const float *dot, const float *sq, int *idx,
                             void *stream) {
  if (b < 0 || n < 0 || k < 1 || k > 100 || k > n) return MVP_EBADSHAPE;
  if (b == 0 || n == 0) return MVP_OK;
  if (!dot || !sq || !idx) return MVP_EBADARG;
  if (b > 65535) return MVP_EBADSHAPE;
  if (k <= kWave && n <= 16384) {
    dim3 wgrid((n + kTwWaves * kTwRows - 1) / (kTwWaves * kTwRows), b);
    const size_t wlds = (size_t)((n + 3) & ~3) * 4;
    if (n % 4 == 0 && (reinterpret_cast<uintptr_t>(dot) & 15) == 0)
      hipLaunchKernelGGL(topk_gram_wave_kernel<4>, wgrid, dim3(kTwWaves * kWave), wlds, as_stream(stream), n, k, dot,
                         sq, idx);
    else
      hipLaunchKernelGGL(topk_gram_wave_kernel<1>, wgrid, dim3(kTwWaves * kWave), wlds, as_stream(stream), n, k, dot,
                         sq, idx);
    return check_launch("mvp_topk_gram");
  }
  const size_t lds = (size_t)(kFkRows * (kFkCols + 1) + kFkCols) * 4 + (size_t)k * kFkRows * 8;
  if (lds > 64 * 1024) return MVP_EBADSHAPE;
  dim3 grid((n + kFkRows - 1) / kFkRows, b);
  hipLaunchKernelGGL(topk_gram_kernel, grid, dim3(kFkRows), lds, as_stream(stream), n, k, dot, sq, idx);
  return check_launch("mvp_topk_gram");
}

extern "C" int mvp_three_nn(int b, int n, int m, const float *unknown,
                            const float *known, float *dist2, int *idx,
                            void *stream) {
  if (b < 0 || n < 0 || m < 0) return MVP_EBADSHAPE;
  if (b == 0 || n == 0) return MVP_OK;
  if (!unknown || !dist2 || !idx || (m > 0 && !known)) return MVP_EBADARG;
  if (b > 65535) return MVP_EBADSHAPE;
  dim3 grid((n + kNnThreads - 1) / kNnThreads, b);
  hipLaunchKernelGGL(three_nn_kernel, grid, dim3(kNnThreads), 0,
                     as_stream(stream), n, m, unknown, known, dist2, idx);
  return check_launch("mvp_three_nn");
}
