// gather_points / grouping_operation / three_interpolate (+ gradients) for
// gfx950.  These are the HBM-bound rows of the hot path: a
// gathered read of a (B,C,N) feature tensor and a streaming write.
//
// Replaces gather_points_kernel/_grad_kernel
// (utils/mm3d_pn2/ops/gather_points/src/gather_points_cuda.cu:8-26,51-70),
// group_points_kernel/_grad_kernel
// (ops/group_points/src/group_points_cuda.cu:56-79,10-31) and
// three_interpolate_kernel/_grad_kernel
// (ops/interpolate/src/three_interpolate_cuda.cu:11-35,61-84).
//
// MI355X-first layout of the work: the reference launches one thread per
// (b, c, output point) and re-reads idx (and the 3 weights) for every channel.
// Here a lane owns one output point, reads its index/weights ONCE into
// registers and walks a strip of kChan channels, so the index traffic is paid
// once per strip instead of once per channel, the gathered reads of one strip
// stay inside one (N x kChan) slab of L2, and the stores are 256-lane
// coalesced rows.  grouping_operation is the same kernel with the
// (npoint, nsample) index array flattened.
#include "common.h"

namespace mvp {

constexpr int kGThreads = 256;
constexpr int kChan = 8;  // channels per workgroup strip

__global__ __launch_bounds__(kGThreads) void gather_kernel(
    int c, int n, int m, const float *__restrict__ points,
    const int *__restrict__ idx, float *__restrict__ out) {
  const int p = blockIdx.x * kGThreads + threadIdx.x;
  if (p >= m) return;
  const int cloud = blockIdx.z;
  const int c0 = blockIdx.y * kChan;
  const int id = idx[(size_t)cloud * m + p];
  const float *src = points + ((size_t)cloud * c + c0) * n + id;
  float *dst = out + ((size_t)cloud * c + c0) * m + p;
  const int cn = min(kChan, c - c0);
  if (cn == kChan) {
    float v[kChan];
#pragma unroll
    for (int k = 0; k < kChan; ++k) v[k] = src[(size_t)k * n];
#pragma unroll
    for (int k = 0; k < kChan; ++k) dst[(size_t)k * m] = v[k];
  } else {
    for (int k = 0; k < cn; ++k) dst[(size_t)k * m] = src[(size_t)k * n];
  }
}

__global__ __launch_bounds__(kGThreads) void gather_grad_kernel(
    int c, int n, int m, const float *__restrict__ grad_out,
    const int *__restrict__ idx, float *__restrict__ grad_points) {
  const int p = blockIdx.x * kGThreads + threadIdx.x;
  if (p >= m) return;
  const int cloud = blockIdx.z;
  const int c0 = blockIdx.y * kChan;
  const int id = idx[(size_t)cloud * m + p];
  const float *src = grad_out + ((size_t)cloud * c + c0) * m + p;
  float *dst = grad_points + ((size_t)cloud * c + c0) * n + id;
  const int cn = min(kChan, c - c0);
  for (int k = 0; k < cn; ++k) atomicAdd(dst + (size_t)k * n, src[(size_t)k * m]);
}

__global__ __launch_bounds__(kGThreads) void three_interpolate_kernel(
    int c, int m, int n, const float *__restrict__ points,
    const int *__restrict__ idx, const float *__restrict__ weight,
    float *__restrict__ out) {
  const int p = blockIdx.x * kGThreads + threadIdx.x;
  if (p >= n) return;
  const int cloud = blockIdx.z;
  const int c0 = blockIdx.y * kChan;
  const int *id = idx + ((size_t)cloud * n + p) * 3;
  const float *w = weight + ((size_t)cloud * n + p) * 3;
  const int i0 = id[0], i1 = id[1], i2 = id[2];
  const float w0 = w[0], w1 = w[1], w2 = w[2];
  const float *src = points + ((size_t)cloud * c + c0) * m;
  float *dst = out + ((size_t)cloud * c + c0) * n + p;
  const int cn = min(kChan, c - c0);
  for (int k = 0; k < cn; ++k) {
    const float *row = src + (size_t)k * m;
    // w0*p0 + w1*p1 + w2*p2 == fma(w2,p2, fma(w1,p1, w0*p0)) (oracle chain)
    dst[(size_t)k * n] =
        __builtin_fmaf(w2, row[i2], __builtin_fmaf(w1, row[i1], w0 * row[i0]));
  }
}

__global__ __launch_bounds__(kGThreads) void three_interpolate_grad_kernel(
    int c, int n, int m, const float *__restrict__ grad_out,
    const int *__restrict__ idx, const float *__restrict__ weight,
    float *__restrict__ grad_points) {
  const int p = blockIdx.x * kGThreads + threadIdx.x;
  if (p >= n) return;
  const int cloud = blockIdx.z;
  const int c0 = blockIdx.y * kChan;
  const int *id = idx + ((size_t)cloud * n + p) * 3;
  const float *w = weight + ((size_t)cloud * n + p) * 3;
  const int i0 = id[0], i1 = id[1], i2 = id[2];
  const float w0 = w[0], w1 = w[1], w2 = w[2];
  const float *src = grad_out + ((size_t)cloud * c + c0) * n + p;
  float *dst = grad_points + ((size_t)cloud * c + c0) * m;
  const int cn = min(kChan, c - c0);
  for (int k = 0; k < cn; ++k) {
    const float g = src[(size_t)k * n];
    float *row = dst + (size_t)k * m;
    atomicAdd(row + i0, g * w0);
    atomicAdd(row + i1, g * w1);
    atomicAdd(row + i2, g * w2);
  }
}

constexpr int kScThreads = 1024;
constexpr int kScFloats = 24576;  // 96 KiB of LDS rows / accumulators per workgroup

// Gather through LDS.  The direct kernel reads 4 scattered bytes per output
// element, i.e. a whole cache sector: with the points of one (cloud, channel)
// row spread over N*4 bytes every block pulls most of that row through L1/L2
// again.  Here a workgroup stages `ch` complete rows of one cloud in LDS with
// coalesced loads (the feature tensor is read exactly once), then produces all
// M outputs of those rows from LDS (random ds_read_b32) with coalesced stores;
// an index is read once per strip.  WEIGHTED: three_interpolate (3 indices and
// weights per output, the oracle's fma chain).
template <bool WEIGHTED>
__global__ __launch_bounds__(kScThreads) void gather_lds_kernel(
    int c, int n_src, int m_out, int ch, const float *__restrict__ points,
    const int *__restrict__ idx, const float *__restrict__ weight, float *__restrict__ out) {
  __shared__ float rows[kScFloats];
  const int t = threadIdx.x;
  const int cloud = blockIdx.y;
  const int c0 = blockIdx.x * ch;
  const int cn = min(ch, c - c0);
  const float *src = points + ((size_t)cloud * c + c0) * n_src;  // rows c0.. are contiguous
  for (int i = t; i < cn * n_src; i += kScThreads) rows[i] = src[i];
  __syncthreads();
  float *dst = out + ((size_t)cloud * c + c0) * m_out;
  if constexpr (!WEIGHTED) {
    const int *id = idx + (size_t)cloud * m_out;
    for (int p = t; p < m_out; p += kScThreads) {
      const int j = id[p];
      for (int k = 0; k < cn; ++k) dst[(size_t)k * m_out + p] = rows[k * n_src + j];
    }
  } else {
    const int *id = idx + (size_t)cloud * m_out * 3;
    const float *w = weight + (size_t)cloud * m_out * 3;
    for (int p = t; p < m_out; p += kScThreads) {
      const int i0 = id[p * 3 + 0], i1 = id[p * 3 + 1], i2 = id[p * 3 + 2];
      const float w0 = w[p * 3 + 0], w1 = w[p * 3 + 1], w2 = w[p * 3 + 2];
      for (int k = 0; k < cn; ++k) {
        const float *row = rows + k * n_src;
        // w0*p0 + w1*p1 + w2*p2 == fma(w2,p2, fma(w1,p1, w0*p0)) (oracle chain)
        dst[(size_t)k * m_out + p] = __builtin_fmaf(w2, row[i2], __builtin_fmaf(w1, row[i1], w0 * row[i0]));
      }
    }
  }
}

// Scatter-add gradients without global atomics.  One workgroup owns `ch`
// channels of one cloud: it accumulates grad_points[b, c0..c0+ch, :] in LDS
// (ds_add_f32) while streaming grad_out rows with coalesced reads -- an index
// is read once and reused for the whole channel strip -- and adds the finished
// rows to the (pre-zeroed) output with coalesced, non-atomic read-modify-
// writes (nobody else touches those rows).  A kNN graph scatters ~k values
// into every source point from spatially close (= temporally close) lanes;
// as global float atomics those collide in L2 (3.5 ms per call at the VRCNet
// shapes), in LDS the whole call is one pass over grad_out.
template <bool WEIGHTED>
__global__ __launch_bounds__(kScThreads) void scatter_lds_kernel(
    int c, int n_dst, int m_src, int ch, const float *__restrict__ grad_out,
    const int *__restrict__ idx, const float *__restrict__ weight,
    float *__restrict__ grad_points) {
  __shared__ float acc[kScFloats];
  const int t = threadIdx.x;
  const int cloud = blockIdx.y;
  const int c0 = blockIdx.x * ch;
  const int cn = min(ch, c - c0);
  for (int i = t; i < cn * n_dst; i += kScThreads) acc[i] = 0.f;
  __syncthreads();
  const float *src = grad_out + ((size_t)cloud * c + c0) * m_src;
  if constexpr (!WEIGHTED) {
    const int *id = idx + (size_t)cloud * m_src;
    for (int p = t; p < m_src; p += kScThreads) {
      const int j = id[p];
      int k = 0;
      for (; k + 4 <= cn; k += 4) {
        const float g0 = src[(size_t)(k + 0) * m_src + p], g1 = src[(size_t)(k + 1) * m_src + p];
        const float g2 = src[(size_t)(k + 2) * m_src + p], g3 = src[(size_t)(k + 3) * m_src + p];
        atomicAdd(&acc[(k + 0) * n_dst + j], g0);
        atomicAdd(&acc[(k + 1) * n_dst + j], g1);
        atomicAdd(&acc[(k + 2) * n_dst + j], g2);
        atomicAdd(&acc[(k + 3) * n_dst + j], g3);
      }
      for (; k < cn; ++k) atomicAdd(&acc[k * n_dst + j], src[(size_t)k * m_src + p]);
    }
  } else {
    const int *id = idx + (size_t)cloud * m_src * 3;
    const float *w = weight + (size_t)cloud * m_src * 3;
    for (int p = t; p < m_src; p += kScThreads) {
      const int i0 = id[p * 3 + 0], i1 = id[p * 3 + 1], i2 = id[p * 3 + 2];
      const float w0 = w[p * 3 + 0], w1 = w[p * 3 + 1], w2 = w[p * 3 + 2];
      for (int k = 0; k < cn; ++k) {
        const float g = src[(size_t)k * m_src + p];
        atomicAdd(&acc[k * n_dst + i0], g * w0);
        atomicAdd(&acc[k * n_dst + i1], g * w1);
        atomicAdd(&acc[k * n_dst + i2], g * w2);
      }
    }
  }
  __syncthreads();
  float *dst = grad_points + ((size_t)cloud * c + c0) * n_dst;  // rows c0.. are contiguous
  for (int i = t; i < cn * n_dst; i += kScThreads) dst[i] += acc[i];
}

// ---------------------------------------------------------------------------
// Scatter-add gradients as a gather: transpose the index list once, then reduce.
//
// The LDS scatter above is bound by ds_add_f32 (the same kernel with plain
// stores is 5.6x faster): every (channel, entry) pair is one LDS atomic, and
// lanes of a wave hit the same destination on kNN graphs.  The destinations of
// an index array do not depend on the channel, so the array is inverted ONCE
// per call -- per cloud and per chunk of grad_out columns a counting sort by
// destination (CSR: offsets + entry list, weights copied next to the entries
// for three_interpolate) -- and every channel strip then owns its destinations
// outright: a thread keeps the sums of its destinations in registers, the
// strip's grad_out columns are staged chunk by chunk in LDS with coalesced loads
// (grad_out is read exactly once) and gathered from there with plain ds_read.
// No atomics on the data path; the per-destination order of the additions is
// the order of the counting sort's fill (like the reference's atomics, not
// fixed).  A chunk is a dependent chain (stage -> barrier -> list -> LDS), so
// the staging buffer is kept at 48 KiB and the kernel at 64 VGPRs: two
// workgroups share a CU and run each other's chains side by side.
constexpr int kTrMaxDst = 8192;   // destinations (counters of the sort live in LDS)
constexpr int kTrFloats = 12288;  // 48 KiB staging buffer: rows x chunk columns
// <= 2048 destinations: 2 per thread, 8 rows x 1536 columns; more: up to 8 per thread, 4 rows x 3072 columns
static int tr_chunk(int n_dst) { return n_dst <= 2 * kScThreads ? 1536 : 3072; }

// Scratch layout, per (cloud, chunk): offsets u16[n_dst + 1] (padded to 8 bytes; a chunk holds at
// most 3072 * 3 entries), then the list: u16 chunk-local column numbers (WEIGHTED: int2 {column,
// weight bits}).  16-bit entries halve what every channel strip re-reads (round 3: the lists and
// offsets were 19 % of the kernel's HBM-side traffic at the VRCNet shapes).
typedef unsigned short tr_u16;
__host__ __device__ inline size_t tr_off_bytes(int n_dst) { return ((size_t)(n_dst + 1) * 2 + 7) & ~(size_t)7; }
__host__ __device__ inline size_t tr_list_bytes(int chunk, int r) { return (size_t)chunk * r * (r == 3 ? 8 : 2); }
__host__ __device__ inline size_t tr_chunk_bytes(int n_dst, int chunk, int r) {
  return (tr_off_bytes(n_dst) + tr_list_bytes(chunk, r) + 7) & ~(size_t)7;
}

// grid (chunks, clouds).
template <bool WEIGHTED>
__global__ __launch_bounds__(kScThreads) void transpose_index_kernel(
    int n_dst, int m_src, int chunk, const int *__restrict__ idx, const float *__restrict__ weight,
    char *__restrict__ scratch) {
  constexpr int R = WEIGHTED ? 3 : 1;
  __shared__ int cnt[kTrMaxDst];
  __shared__ int wtot[kScThreads / kWave];
  const int t = threadIdx.x, lane = t & (kWave - 1), wave = t >> 6;
  const int q = blockIdx.x, cloud = blockIdx.y, nq = gridDim.x;
  const int p0 = q * chunk;
  const int ne = min(chunk, m_src - p0) * R;  // entries of this chunk
  const int *id = idx + ((size_t)cloud * m_src + p0) * R;
  for (int j = t; j < n_dst; j += kScThreads) cnt[j] = 0;
  __syncthreads();
  for (int e = t; e < ne; e += kScThreads) {
    const int j = id[e];
    if ((unsigned)j < (unsigned)n_dst) atomicAdd(&cnt[j], 1);
  }
  __syncthreads();
  // exclusive scan: thread t owns `per` consecutive counters
  const int per = (n_dst + kScThreads - 1) / kScThreads;  // <= 8
  const int j0 = t * per;
  int sum = 0;
  for (int i = 0; i < per; ++i) sum += j0 + i < n_dst ? cnt[j0 + i] : 0;
  int incl = sum;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const int o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == kWave - 1) wtot[wave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wtot[w];
  int run = base + incl - sum;
  char *base_p = scratch + ((size_t)cloud * nq + q) * tr_chunk_bytes(n_dst, chunk, R);
  tr_u16 *off = reinterpret_cast<tr_u16 *>(base_p);
  for (int i = 0; i < per; ++i) {
    if (j0 + i < n_dst) {
      const int cc = cnt[j0 + i];
      cnt[j0 + i] = run;  // becomes the fill cursor
      off[j0 + i] = (tr_u16)run;
      run += cc;
    }
  }
  if (t == kScThreads - 1) off[n_dst] = (tr_u16)run;
  __syncthreads();
  char *lst = base_p + tr_off_bytes(n_dst);
  const float *w = WEIGHTED ? weight + ((size_t)cloud * m_src + p0) * R : nullptr;
  for (int e = t; e < ne; e += kScThreads) {
    const int j = id[e];
    if ((unsigned)j < (unsigned)n_dst) {
      const int pos = atomicAdd(&cnt[j], 1);
      if constexpr (WEIGHTED) {
        reinterpret_cast<int2 *>(lst)[pos] = make_int2(e / 3, __float_as_int(w[e]));
      } else {
        reinterpret_cast<tr_u16 *>(lst)[pos] = (tr_u16)e;
      }
    }
  }
}

// grid (channel strips of CH rows, clouds).  D = destinations per thread,
// CHUNK = columns staged per step (CH * CHUNK floats of LDS).
// OVERWRITE: grad_points is written (every element), not accumulated into -- no zero fill by
// the caller, no read of the destination.
// (Round 3: fetching the NEXT chunk's columns into registers under the current chunk's gather costs the second
// workgroup per CU -- 12-16 more VGPRs -- and loses: 233 -> 278 us at (64,64,3072) <- 49152, profiles/r3_scatter_grad.txt.)
template <bool WEIGHTED, int D, int CH, int CHUNK, bool OVERWRITE>
__global__ __launch_bounds__(kScThreads) __attribute__((amdgpu_waves_per_eu(D == 8 ? 4 : 8, 8))) void transposed_reduce_kernel(
    int c, int n_dst, int m_src, const float *__restrict__ grad_out, const char *__restrict__ scratch,
    float *__restrict__ grad_points) {
  constexpr int R = WEIGHTED ? 3 : 1;
  static_assert(CH * CHUNK <= kTrFloats, "staging buffer");
  __shared__ float stage[CH * CHUNK];
  const int t = threadIdx.x;
  const int cloud = blockIdx.y;
  const int c0 = blockIdx.x * CH;
  const int cn = min(CH, c - c0);
  const int nq = (m_src + CHUNK - 1) / CHUNK;
  const float *src = grad_out + ((size_t)cloud * c + c0) * m_src;
  float acc[D][CH];
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int k = 0; k < CH; ++k) acc[d][k] = 0.f;
  for (int q = 0; q < nq; ++q) {
    const int p0 = q * CHUNK;
    const int pn = min(CHUNK, m_src - p0);
    const char *base_p = scratch + ((size_t)cloud * nq + q) * tr_chunk_bytes(n_dst, CHUNK, R);
    const tr_u16 *off = reinterpret_cast<const tr_u16 *>(base_p);
    const char *lst = base_p + tr_off_bytes(n_dst);
    // List reads are unconditional (position clamped into the chunk's list, the
    // result discarded when !live): four of them issue back to back instead of
    // four exec-masked load + wait sequences.
    auto entry = [&](int i, bool live, int &e, float &w) {  // list position i -> chunk-local column, weight (0 if !live)
      const int ii = min(i, CHUNK * R - 1);
      if constexpr (WEIGHTED) {
        const int2 ew = reinterpret_cast<const int2 *>(lst)[ii];
        e = live ? ew.x : 0;
        w = live ? __int_as_float(ew.y) : 0.f;
      } else {
        const int ev = reinterpret_cast<const tr_u16 *>(lst)[ii];
        e = live ? ev : 0;
        w = live ? 1.f : 0.f;
      }
    };
    // offsets and the first entry of every destination are fetched under the
    // staging loads (a kNN graph has about one entry per destination and chunk)
    int o0[D], o1[D], e0[D];
    float w0[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int j = t + d * kScThreads;
      o0[d] = j < n_dst ? off[j] : 0;
      o1[d] = j < n_dst ? off[j + 1] : 0;
    }
#pragma unroll
    for (int d = 0; d < D; ++d) entry(o0[d], o0[d] < o1[d], e0[d], w0[d]);
    __syncthreads();  // the previous chunk is no longer read
    // all loads of a column block are issued before the first LDS store
#pragma unroll
    for (int i0 = 0; i0 < CHUNK; i0 += kScThreads) {
      const int i = i0 + t;
      float v[CH];
#pragma unroll
      for (int k = 0; k < CH; ++k) v[k] = (k < cn && i < pn) ? src[(size_t)k * m_src + p0 + i] : 0.f;
      if (i < CHUNK) {
#pragma unroll
        for (int k = 0; k < CH; ++k) stage[k * CHUNK + i] = v[k];
      }
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (o0[d] < o1[d]) {
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          const float g = stage[k * CHUNK + e0[d]];
          acc[d][k] += WEIGHTED ? g * w0[d] : g;
        }
      }
      for (int i = o0[d] + 1; i < o1[d]; i += 4) {  // four independent list reads per step
        int e[4];
        float w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) entry(i + u, i + u < o1[d], e[u], w[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool live = i + u < o1[d];
#pragma unroll
          for (int k = 0; k < CH; ++k) {
            const float g = stage[k * CHUNK + e[u]];  // column 0 when !live: read, not added
            const float term = WEIGHTED ? g * w[u] : g;
            acc[d][k] += live ? term : 0.f;
          }
        }
      }
    }
  }
  float *dst = grad_points + ((size_t)cloud * c + c0) * n_dst;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const int j = t + d * kScThreads;
    if (j < n_dst) {
#pragma unroll
      for (int k = 0; k < CH; ++k)
        if (k < cn) {
          if constexpr (OVERWRITE) dst[(size_t)k * n_dst + j] = acc[d][k];
          else dst[(size_t)k * n_dst + j] += acc[d][k];
        }
    }
  }
}

// ---------------------------------------------------------------------------
// Gather + max over the neighbours of a point, fused (widening row N2, the pooling half):
//   out[b,c,p] = max_j points[b,c, idx[b,p,j]],   arg[b,c,p] = idx[b,p,j*], j* the FIRST maximal j
// edge_preserve_sampling (completion/model_utils.py:88-110 of the reference) gathers the
// (B, C, P, k) neighbour tensor and reduces it with torch.max: at VRCNet's first level that is
// 0.8 GB written, read again, and the same twice more on the way back.  Here a workgroup stages
// `ch` complete rows of one cloud in LDS (the feature tensor is read exactly once, as in
// gather_lds_kernel), a thread owns one output point, reads its k indices once per strip and keeps
// the running maxima of the strip's channels in registers: the neighbour tensor never exists.
// The winner's SOURCE index is kept for the gradient, which then is a scatter of P values per row
// (gather_max_grad_kernel) instead of a pass over P*k of them.
constexpr int kGmChan = 16;   // rows per strip at most (register arrays)
__global__ __launch_bounds__(kScThreads) void gather_max_lds_kernel(
    int c, int n_src, int p_out, int k, int ch, const float *__restrict__ points,
    const int *__restrict__ idx, float *__restrict__ out, int *__restrict__ arg) {
  __shared__ float rows[kScFloats];
  const int t = threadIdx.x;
  const int cloud = blockIdx.y;
  const int c0 = blockIdx.x * ch;
  const int cn = min(ch, c - c0);
  const float *src = points + ((size_t)cloud * c + c0) * n_src;  // rows c0.. are contiguous
  for (int i = t; i < cn * n_src; i += kScThreads) rows[i] = src[i];
  __syncthreads();
  float *dst = out + ((size_t)cloud * c + c0) * p_out;
  int *adst = arg + ((size_t)cloud * c + c0) * p_out;
  const int *id = idx + (size_t)cloud * p_out * k;
  for (int p = t; p < p_out; p += kScThreads) {
    float best[kGmChan];
    int bi[kGmChan];
#pragma unroll
    for (int q = 0; q < kGmChan; ++q) {
      best[q] = -__builtin_inff();
      bi[q] = 0;
    }
    for (int j = 0; j < k; ++j) {
      // (an index outside the cloud would be an LDS read outside the strip: clamped, as no valid input has one)
      const int s = min(max(id[(size_t)p * k + j], 0), n_src - 1);
      if (j == 0) {
#pragma unroll
        for (int q = 0; q < kGmChan; ++q) bi[q] = s;   // a row of -inf keeps the first neighbour
      }
#pragma unroll
      for (int q = 0; q < kGmChan; ++q) {
        if (q < cn) {
          const float v = rows[q * n_src + s];
          // strict: the first maximum wins, and the first NaN wins for good (torch.max's rules: a diverging
          // run shows its NaN at the pooling layer, as on the gather_points + torch.max route)
          const bool gt = v > best[q] || (v != v && best[q] == best[q]);
          best[q] = gt ? v : best[q];
          bi[q] = gt ? s : bi[q];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < kGmChan; ++q) {
      if (q < cn) {
        dst[(size_t)q * p_out + p] = best[q];
        adst[(size_t)q * p_out + p] = bi[q];
      }
    }
  }
}

// grad_points[b,c, arg[b,c,p]] += grad_out[b,c,p]: one workgroup owns `ch` rows of one cloud and
// accumulates them in LDS (ds_add_f32: several output points may share a winner), then writes
// (OVERWRITE) or adds the finished rows with coalesced stores.
template <bool OVERWRITE>
__global__ __launch_bounds__(kScThreads) void gather_max_grad_kernel(
    int c, int n_dst, int p_src, int ch, const float *__restrict__ grad_out, const int *__restrict__ arg,
    float *__restrict__ grad_points) {
  __shared__ float acc[kScFloats];
  const int t = threadIdx.x;
  const int cloud = blockIdx.y;
  const int c0 = blockIdx.x * ch;
  const int cn = min(ch, c - c0);
  for (int i = t; i < cn * n_dst; i += kScThreads) acc[i] = 0.f;
  __syncthreads();
  const float *src = grad_out + ((size_t)cloud * c + c0) * p_src;
  const int *asrc = arg + ((size_t)cloud * c + c0) * p_src;
  for (int i = t; i < cn * p_src; i += kScThreads) {   // (row q, point p) = (i / p_src, i % p_src): coalesced
    const int q = i / p_src;
    const int j = asrc[i];
    if ((unsigned)j < (unsigned)n_dst) atomicAdd(&acc[q * n_dst + j], src[i]);
  }
  __syncthreads();
  float *dst = grad_points + ((size_t)cloud * c + c0) * n_dst;
  for (int i = t; i < cn * n_dst; i += kScThreads) {
    if constexpr (OVERWRITE) dst[i] = acc[i];
    else dst[i] += acc[i];
  }
}

static long long transposed_scratch_bytes(int b, int n_dst, int m_src, int r) {
  if (b <= 0 || b > 65535 || n_dst <= 0 || m_src <= 0 || (r != 1 && r != 3)) return 0;
  if (n_dst > kTrMaxDst) return 0;
  const int chunk = tr_chunk(n_dst);
  const long long nq = (m_src + chunk - 1) / chunk;
  if (nq > 65535) return 0;
  return (long long)b * nq * (long long)tr_chunk_bytes(n_dst, chunk, r);
}

// mode bit 0 (MVP_SCATTER_OVERWRITE): write instead of accumulate; bit 1 (MVP_SCATTER_INDEX_READY):
// the scratch already holds the transposed index of exactly this (idx, weight) -- skip the sort.
template <bool WEIGHTED>
static void transposed_scatter(int b, int c, int n_dst, int m_src, const float *grad_out, const int *idx,
                               const float *weight, float *grad_points, void *scratch, int mode, hipStream_t stream) {
  const int chunk = tr_chunk(n_dst);
  const int nq = (m_src + chunk - 1) / chunk;
  char *sbytes = static_cast<char *>(scratch);
  if (!(mode & MVP_SCATTER_INDEX_READY))
    hipLaunchKernelGGL(transpose_index_kernel<WEIGHTED>, dim3(nq, b), dim3(kScThreads), 0, stream, n_dst, m_src, chunk,
                       idx, weight, sbytes);
#define MVP_TR_LAUNCH(DD, CH, CHUNK)                                                                              \
  do {                                                                                                            \
    if (mode & MVP_SCATTER_OVERWRITE)                                                                             \
      hipLaunchKernelGGL((transposed_reduce_kernel<WEIGHTED, DD, CH, CHUNK, true>), dim3((c + CH - 1) / CH, b),   \
                         dim3(kScThreads), 0, stream, c, n_dst, m_src, grad_out, sbytes, grad_points);            \
    else                                                                                                          \
      hipLaunchKernelGGL((transposed_reduce_kernel<WEIGHTED, DD, CH, CHUNK, false>), dim3((c + CH - 1) / CH, b),  \
                         dim3(kScThreads), 0, stream, c, n_dst, m_src, grad_out, sbytes, grad_points);            \
  } while (0)
  if (n_dst <= kScThreads) MVP_TR_LAUNCH(1, 8, 1536);
  else if (n_dst <= 2 * kScThreads) MVP_TR_LAUNCH(2, 8, 1536);
  else if (n_dst <= 4 * kScThreads) MVP_TR_LAUNCH(4, 4, 3072);
  else MVP_TR_LAUNCH(8, 4, 3072);
#undef MVP_TR_LAUNCH
}

// channels per workgroup for the LDS scatter; 0 = rows too long, use atomics
static int scatter_channels(int c, int n_dst) {
  int ch = kScFloats / (n_dst > 0 ? n_dst : 1);
  if (ch > c) ch = c;
  if (ch > 16) ch = 16;
  return ch;
}

static bool grid_ok(long long gx, long long gy, long long gz) {
  return gx <= 2147483647LL && gy <= 65535 && gz <= 65535;
}

}  // namespace mvp

using namespace mvp;

extern "C" int mvp_gather_points(int b, int c, int n, int npoints,
                                 const float *points, const int *idx,
                                 float *out, void *stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0) return MVP_EBADSHAPE;
  if (b == 0 || c == 0 || npoints == 0) return MVP_OK;
  if (n == 0) return MVP_EBADSHAPE;
  if (!points || !idx || !out) return MVP_EBADARG;
  // rows short enough to stage and enough outputs per row to pay for staging
  if (const int ch = scatter_channels(c, n); ch > 0 && 2LL * npoints >= n) {
    dim3 sgrid((c + ch - 1) / ch, b);
    if (!grid_ok(sgrid.x, sgrid.y, 1)) return MVP_EBADSHAPE;
    hipLaunchKernelGGL(gather_lds_kernel<false>, sgrid, dim3(kScThreads), 0, as_stream(stream), c, n, npoints,
                       ch, points, idx, static_cast<const float *>(nullptr), out);
    return check_launch("mvp_gather_points");
  }
  dim3 grid((npoints + kGThreads - 1) / kGThreads, (c + kChan - 1) / kChan, b);
  if (!grid_ok(grid.x, grid.y, grid.z)) return MVP_EBADSHAPE;
  hipLaunchKernelGGL(gather_kernel, grid, dim3(kGThreads), 0, as_stream(stream),
                     c, n, npoints, points, idx, out);
  return check_launch("mvp_gather_points");
}

extern "C" int mvp_gather_points_grad(int b, int c, int n, int npoints,
                                      const float *grad_out, const int *idx,
                                      float *grad_points, void *stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0) return MVP_EBADSHAPE;
  if (b == 0 || c == 0 || npoints == 0) return MVP_OK;
  if (n == 0) return MVP_EBADSHAPE;
  if (!grad_out || !idx || !grad_points) return MVP_EBADARG;
  if (const int ch = scatter_channels(c, n)) {
    dim3 sgrid((c + ch - 1) / ch, b);
    if (!grid_ok(sgrid.x, sgrid.y, 1)) return MVP_EBADSHAPE;
    hipLaunchKernelGGL(scatter_lds_kernel<false>, sgrid, dim3(kScThreads), 0, as_stream(stream), c, n,
                       npoints, ch, grad_out, idx, static_cast<const float *>(nullptr), grad_points);
    return check_launch("mvp_gather_points_grad");
  }
  dim3 grid((npoints + kGThreads - 1) / kGThreads, (c + kChan - 1) / kChan, b);
  if (!grid_ok(grid.x, grid.y, grid.z)) return MVP_EBADSHAPE;
  hipLaunchKernelGGL(gather_grad_kernel, grid, dim3(kGThreads), 0,
                     as_stream(stream), c, n, npoints, grad_out, idx,
                     grad_points);
  return check_launch("mvp_gather_points_grad");
}

extern "C" int mvp_group_points(int b, int c, int n, int npoints, int nsample,
                                const float *points, const int *idx, float *out,
                                void *stream) {
  if (npoints < 0 || nsample < 0) return MVP_EBADSHAPE;
  const long long flat = (long long)npoints * nsample;
  if (flat > 2147483647LL) return MVP_EBADSHAPE;
  return mvp_gather_points(b, c, n, (int)flat, points, idx, out, stream);
}

extern "C" int mvp_group_points_grad(int b, int c, int n, int npoints,
                                     int nsample, const float *grad_out,
                                     const int *idx, float *grad_points,
                                     void *stream) {
  if (npoints < 0 || nsample < 0) return MVP_EBADSHAPE;
  const long long flat = (long long)npoints * nsample;
  if (flat > 2147483647LL) return MVP_EBADSHAPE;
  return mvp_gather_points_grad(b, c, n, (int)flat, grad_out, idx, grad_points,
                                stream);
}

extern "C" int mvp_three_interpolate(int b, int c, int m, int n,
                                     const float *points, const int *idx,
                                     const float *weight, float *out,
                                     void *stream) {
  if (b < 0 || c < 0 || m < 0 || n < 0) return MVP_EBADSHAPE;
  if (b == 0 || c == 0 || n == 0) return MVP_OK;
  if (m == 0) return MVP_EBADSHAPE;
  if (!points || !idx || !weight || !out) return MVP_EBADARG;
  if (const int ch = scatter_channels(c, m); ch > 0 && 2LL * n >= m) {
    dim3 sgrid((c + ch - 1) / ch, b);
    if (!grid_ok(sgrid.x, sgrid.y, 1)) return MVP_EBADSHAPE;
    hipLaunchKernelGGL(gather_lds_kernel<true>, sgrid, dim3(kScThreads), 0, as_stream(stream), c, m, n, ch,
                       points, idx, weight, out);
    return check_launch("mvp_three_interpolate");
  }
  dim3 grid((n + kGThreads - 1) / kGThreads, (c + kChan - 1) / kChan, b);
  if (!grid_ok(grid.x, grid.y, grid.z)) return MVP_EBADSHAPE;
  hipLaunchKernelGGL(three_interpolate_kernel, grid, dim3(kGThreads), 0,
                     as_stream(stream), c, m, n, points, idx, weight, out);
  return check_launch("mvp_three_interpolate");
}

extern "C" int mvp_three_interpolate_grad(int b, int c, int n, int m,
                                          const float *grad_out, const int *idx,
                                          const float *weight,
                                          float *grad_points, void *stream) {
  if (b < 0 || c < 0 || m < 0 || n < 0) return MVP_EBADSHAPE;
  if (b == 0 || c == 0 || n == 0) return MVP_OK;
  if (m == 0) return MVP_EBADSHAPE;
  if (!grad_out || !idx || !weight || !grad_points) return MVP_EBADARG;
  if (const int ch = scatter_channels(c, m)) {
    dim3 sgrid((c + ch - 1) / ch, b);
    if (!grid_ok(sgrid.x, sgrid.y, 1)) return MVP_EBADSHAPE;
    hipLaunchKernelGGL(scatter_lds_kernel<true>, sgrid, dim3(kScThreads), 0, as_stream(stream), c, m, n,
                       ch, grad_out, idx, weight, grad_points);
    return check_launch("mvp_three_interpolate_grad");
  }
  dim3 grid((n + kGThreads - 1) / kGThreads, (c + kChan - 1) / kChan, b);
  if (!grid_ok(grid.x, grid.y, grid.z)) return MVP_EBADSHAPE;
  hipLaunchKernelGGL(three_interpolate_grad_kernel, grid, dim3(kGThreads), 0,
                     as_stream(stream), c, n, m, grad_out, idx, weight,
                     grad_points);
  return check_launch("mvp_three_interpolate_grad");
}

// ---- gradients with caller-provided scratch (transposed index list)
extern "C" long long mvp_scatter_scratch_bytes(int b, int n_dst, int m_src, int r) {
  return transposed_scratch_bytes(b, n_dst, m_src, r);
}

// The plain kernels accumulate: an overwrite request that falls back to them zero-fills first.
static int zero_for_overwrite(int mode, float *grad_points, long long count, void *stream) {
  if (!(mode & MVP_SCATTER_OVERWRITE) || count <= 0) return MVP_OK;
  if (!grad_points) return MVP_EBADARG;
  if (hipMemsetAsync(grad_points, 0, (size_t)count * sizeof(float), as_stream(stream)) != hipSuccess)
    return check_launch("mvp_*_grad_ws (zero fill)");
  return MVP_OK;
}

extern "C" int mvp_gather_points_grad_ws(int b, int c, int n, int npoints, const float *grad_out, const int *idx,
                                         float *grad_points, void *scratch, long long scratch_bytes, int mode,
                                         void *stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0) return MVP_EBADSHAPE;
  const long long need = transposed_scratch_bytes(b, n, npoints, 1);
  if (c == 0 || npoints == 0 || !scratch || need == 0 || scratch_bytes < need) {
    if (const int rc = zero_for_overwrite(mode, grad_points, (long long)b * c * n, stream)) return rc;
    return mvp_gather_points_grad(b, c, n, npoints, grad_out, idx, grad_points, stream);
  }
  if (!grad_out || !idx || !grad_points) return MVP_EBADARG;
  transposed_scatter<false>(b, c, n, npoints, grad_out, idx, nullptr, grad_points, scratch, mode, as_stream(stream));
  return check_launch("mvp_gather_points_grad_ws");
}

extern "C" int mvp_group_points_grad_ws(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                        const int *idx, float *grad_points, void *scratch,
                                        long long scratch_bytes, int mode, void *stream) {
  if (npoints < 0 || nsample < 0) return MVP_EBADSHAPE;
  const long long flat = (long long)npoints * nsample;
  if (flat > 2147483647LL) return MVP_EBADSHAPE;
  return mvp_gather_points_grad_ws(b, c, n, (int)flat, grad_out, idx, grad_points, scratch, scratch_bytes, mode,
                                   stream);
}

extern "C" int mvp_three_interpolate_grad_ws(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                             const float *weight, float *grad_points, void *scratch,
                                             long long scratch_bytes, int mode, void *stream) {
  if (b < 0 || c < 0 || m < 0 || n < 0) return MVP_EBADSHAPE;
  const long long need = transposed_scratch_bytes(b, m, n, 3);
  if (c == 0 || n == 0 || !scratch || need == 0 || scratch_bytes < need) {
    if (const int rc = zero_for_overwrite(mode, grad_points, (long long)b * c * m, stream)) return rc;
    return mvp_three_interpolate_grad(b, c, n, m, grad_out, idx, weight, grad_points, stream);
  }
  if (!grad_out || !idx || !weight || !grad_points) return MVP_EBADARG;
  transposed_scatter<true>(b, c, m, n, grad_out, idx, weight, grad_points, scratch, mode, as_stream(stream));
  return check_launch("mvp_three_interpolate_grad_ws");
}

extern "C" int mvp_gather_max(int b, int c, int n, int npoints, int k, const float *points, const int *idx,
                              float *out, int *arg, void *stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0 || k <= 0) return MVP_EBADSHAPE;
  if (b == 0 || c == 0 || npoints == 0) return MVP_OK;
  if (n == 0 || b > 65535) return MVP_EBADSHAPE;
  if (!points || !idx || !out || !arg) return MVP_EBADARG;
  const int ch = min(scatter_channels(c, n), kGmChan);
  if (ch <= 0) return MVP_EBADSHAPE;   // rows longer than the LDS staging buffer: the caller gathers + reduces
  dim3 grid((c + ch - 1) / ch, b);
  if (!grid_ok(grid.x, grid.y, 1)) return MVP_EBADSHAPE;
  hipLaunchKernelGGL(gather_max_lds_kernel, grid, dim3(kScThreads), 0, as_stream(stream), c, n, npoints, k, ch,
                     points, idx, out, arg);
  return check_launch("mvp_gather_max");
}

extern "C" int mvp_gather_max_grad(int b, int c, int n, int npoints, const float *grad_out, const int *arg,
                                   float *grad_points, int overwrite, void *stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0) return MVP_EBADSHAPE;
  if (b == 0 || c == 0 || n == 0) return MVP_OK;
  if (b > 65535) return MVP_EBADSHAPE;
  if (!grad_out || !arg || !grad_points) return MVP_EBADARG;
  const int ch = scatter_channels(c, n);
  if (ch <= 0) return MVP_EBADSHAPE;
  dim3 grid((c + ch - 1) / ch, b);
  if (!grid_ok(grid.x, grid.y, 1)) return MVP_EBADSHAPE;
  if (overwrite)
    hipLaunchKernelGGL(gather_max_grad_kernel<true>, grid, dim3(kScThreads), 0, as_stream(stream), c, n, npoints, ch,
                       grad_out, arg, grad_points);
  else
    hipLaunchKernelGGL(gather_max_grad_kernel<false>, grid, dim3(kScThreads), 0, as_stream(stream), c, n, npoints, ch,
                       grad_out, arg, grad_points);
  return check_launch("mvp_gather_max_grad");
}
