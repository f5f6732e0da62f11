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
