// Shared-weight neighbourhood aggregation of VRCNet's point self-attention
// (the grouped-feature rows): per point, the k neighbours' value
// vectors are summed with per-neighbour weights that are shared by `share`
// channel groups,
//   out[b, s*cw + m, n] = sum_k w[b, m, k, n] * v[b, s*cw + m, k, n].
// The reference has no kernel for it: completion/models/vrcnet.py:52-55
// materialises `w.repeat(1, share, 1, 1)`, the product and reduces it
// (three passes over the (B, C, k, N) tensor forward, five backward).  Here it is
// one streaming pass forward and one backward: a lane owns one (b, m, n), keeps
// the `share` sums in registers and reads every weight once for all groups; all
// accesses are coalesced along n.  Pure HBM streaming: 4 (share + 1) cw k n bytes
// read, 4 share cw n written per cloud.
#include "common.h"

namespace mvp {

constexpr int kAgThreads = 256;

template <int SHARE>
__global__ __launch_bounds__(kAgThreads) void share_weighted_sum_kernel(
    int cw, int k, int n, const float *__restrict__ w, const float *__restrict__ v, float *__restrict__ out) {
  const int p = blockIdx.x * kAgThreads + threadIdx.x;
  if (p >= n) return;
  const int m = blockIdx.y, cloud = blockIdx.z;
  const size_t kn = (size_t)k * n;
  const float *wp = w + ((size_t)cloud * cw + m) * kn + p;
  const float *vp = v + ((size_t)cloud * SHARE * cw + m) * kn + p;
  float acc[SHARE];
#pragma unroll
  for (int s = 0; s < SHARE; ++s) acc[s] = 0.f;
#pragma unroll 2
  for (int kk = 0; kk < k; ++kk) {
    const float wk = wp[(size_t)kk * n];
#pragma unroll
    for (int s = 0; s < SHARE; ++s) acc[s] += wk * vp[(size_t)s * cw * kn + (size_t)kk * n];  // k ascending, mul then add
  }
  float *op = out + ((size_t)cloud * SHARE * cw + m) * n + p;
#pragma unroll
  for (int s = 0; s < SHARE; ++s) op[(size_t)s * cw * n] = acc[s];
}

// grad_v[b, s*cw+m, k, n] = w[b,m,k,n] * g[b, s*cw+m, n];
// grad_w[b, m, k, n] = sum_s g[b, s*cw+m, n] * v[b, s*cw+m, k, n]   (s ascending)
template <int SHARE>
__global__ __launch_bounds__(kAgThreads) void share_weighted_sum_grad_kernel(
    int cw, int k, int n, const float *__restrict__ w, const float *__restrict__ v,
    const float *__restrict__ grad_out, float *__restrict__ grad_w, float *__restrict__ grad_v) {
  const int p = blockIdx.x * kAgThreads + threadIdx.x;
  if (p >= n) return;
  const int m = blockIdx.y, cloud = blockIdx.z;
  const size_t kn = (size_t)k * n;
  const size_t wo = ((size_t)cloud * cw + m) * kn + p;
  const size_t vo = ((size_t)cloud * SHARE * cw + m) * kn + p;
  const float *gp = grad_out + ((size_t)cloud * SHARE * cw + m) * n + p;
  float g[SHARE];
#pragma unroll
  for (int s = 0; s < SHARE; ++s) g[s] = gp[(size_t)s * cw * n];
#pragma unroll 2
  for (int kk = 0; kk < k; ++kk) {
    const float wk = w[wo + (size_t)kk * n];
    float gw = 0.f;
#pragma unroll
    for (int s = 0; s < SHARE; ++s) {
      const size_t o = vo + (size_t)s * cw * kn + (size_t)kk * n;
      gw += g[s] * v[o];
      grad_v[o] = wk * g[s];
    }
    grad_w[wo + (size_t)kk * n] = gw;
  }
}

// The same aggregation with the GATHER of the neighbours' values fused in (round 4):
//   out[b, s*cw + m, p] = sum_k w[b, m, k, p] * v[b, s*cw + m, idx[b, k, p]]
// v (B, C, n_src) are the per-point values, idx (B, k, n) the neighbour lists (k-major, as the models'
// get_edge_features hands them to the grouping operator).  The (B, C, k, n) tensor of gathered values -- 252 MB per
// SA_module of VRCNet at every level, written by the grouping kernel, read by the kernel above, read again by its
// gradient -- is never formed: a workgroup stages the `share` rows of one weight channel m in LDS, interleaved by point
// ({v_0[j], .., v_{share-1}[j]}: one 16-byte read per four groups), and a lane owns a position p.  Same arithmetic,
// same order as mvp_group_points + share_weighted_sum_kernel: bit-identical results.
// GRAD: grad_w[b, m, k, p] = sum_s g[b, s*cw+m, p] * v[b, s*cw+m, idx[b,k,p]]   (s ascending), and the gradient of the
// gathered values, grad_vals[b, s*cw+m, k, p] = w[b,m,k,p] * g[b, s*cw+m, p], which the grouping operator's backward
// (mvp_group_points_grad_ws: inverted index, no atomics) scatters into grad_v.
constexpr int kGsThreads = 1024;
constexpr int kGsMaxLds = 96 * 1024;

template <int SHARE, bool GRAD>
__global__ __launch_bounds__(kGsThreads) void share_gather_sum_kernel(
    int cw, int k, int n_src, int n, int chunk, const float *__restrict__ w, const float *__restrict__ v,
    const int *__restrict__ idx, const float *__restrict__ g, float *__restrict__ out, float *__restrict__ grad_w,
    float *__restrict__ grad_vals) {
  __shared__ __attribute__((aligned(16))) float s_rows[kGsMaxLds / 4];   // [n_src][SHARE]
  const int t = threadIdx.x;
  const int m = blockIdx.y, cloud = blockIdx.z;
  const size_t crow = (size_t)cloud * SHARE * cw + m;              // row of group 0; group s: + s * cw
#pragma unroll
  for (int s = 0; s < SHARE; ++s)
    for (int j = t; j < n_src; j += kGsThreads) s_rows[(size_t)j * SHARE + s] = v[(crow + (size_t)s * cw) * n_src + j];
  __syncthreads();
  const int p_end = min(n, (int)(blockIdx.x + 1) * chunk);
  const size_t kn = (size_t)k * n;
  for (int p = blockIdx.x * chunk + t; p < p_end; p += kGsThreads) {
    const int *ip = idx + (size_t)cloud * kn + p;
    const float *wp = w + ((size_t)cloud * cw + m) * kn + p;
    float acc[SHARE];
#pragma unroll
    for (int s = 0; s < SHARE; ++s) acc[s] = GRAD ? g[(crow + (size_t)s * cw) * n + p] : 0.f;   // GRAD: the groups' grad_out
    for (int kk = 0; kk < k; ++kk) {
      const int j = min(max(ip[(size_t)kk * n], 0), n_src - 1);
      const float wk = wp[(size_t)kk * n];
      const float *row = s_rows + (size_t)j * SHARE;
      if constexpr (GRAD) {
        float gw = 0.f;
#pragma unroll
        for (int s = 0; s < SHARE; ++s) {
          gw += acc[s] * row[s];
          grad_vals[((crow + (size_t)s * cw) * k + kk) * n + p] = wk * acc[s];
        }
        grad_w[((size_t)cloud * cw + m) * kn + (size_t)kk * n + p] = gw;
      } else {
#pragma unroll
        for (int s = 0; s < SHARE; ++s) acc[s] += wk * row[s];     // k ascending, mul then add
      }
    }
    if constexpr (!GRAD) {
#pragma unroll
      for (int s = 0; s < SHARE; ++s) out[(crow + (size_t)s * cw) * n + p] = acc[s];
    }
  }
}

static bool ag_shape_ok(int b, int share, int cw, int k, int n) {
  return b >= 0 && cw >= 0 && k >= 0 && n >= 0 && (share == 1 || share == 2 || share == 4 || share == 8 || share == 16) &&
         b <= 65535 && cw <= 65535;
}

}  // namespace mvp

using namespace mvp;

#define MVP_AG_DISPATCH(KERNEL, ...)                                                         \
  switch (share) {                                                                           \
    case 1: hipLaunchKernelGGL(KERNEL<1>, grid, dim3(kAgThreads), 0, st, __VA_ARGS__); break;   \
    case 2: hipLaunchKernelGGL(KERNEL<2>, grid, dim3(kAgThreads), 0, st, __VA_ARGS__); break;   \
    case 4: hipLaunchKernelGGL(KERNEL<4>, grid, dim3(kAgThreads), 0, st, __VA_ARGS__); break;   \
    case 8: hipLaunchKernelGGL(KERNEL<8>, grid, dim3(kAgThreads), 0, st, __VA_ARGS__); break;   \
    default: hipLaunchKernelGGL(KERNEL<16>, grid, dim3(kAgThreads), 0, st, __VA_ARGS__); break; \
  }

extern "C" int mvp_share_weighted_sum(int b, int share, int cw, int k, int n, const float *w, const float *v,
                                      float *out, void *stream) {
  if (!ag_shape_ok(b, share, cw, k, n)) return MVP_EBADSHAPE;
  if (b == 0 || cw == 0 || n == 0) return MVP_OK;
  if (!out || (k > 0 && (!w || !v))) return MVP_EBADARG;
  const dim3 grid((n + kAgThreads - 1) / kAgThreads, cw, b);
  hipStream_t st = as_stream(stream);
  MVP_AG_DISPATCH(share_weighted_sum_kernel, cw, k, n, w, v, out)
  return check_launch("mvp_share_weighted_sum");
}

// positions per workgroup: enough workgroups for the chip (each stages its `share` rows again)
static int gs_chunk(int b, int cw, int n) {
  int parts = 1;
  while ((long long)b * cw * parts < 512 && n / (parts * 2) >= 256) parts *= 2;
  return (n + parts - 1) / parts;
}

extern "C" long long mvp_share_gather_sum_lds_bytes(int share, int n_src) { return (long long)share * n_src * 4; }

extern "C" int mvp_share_gather_sum(int b, int share, int cw, int k, int n_src, int n, const float *w, const float *v,
                                    const int *idx, float *out, void *stream) {
  if (!ag_shape_ok(b, share, cw, k, n) || n_src <= 0 || (long long)share * n_src * 4 > kGsMaxLds) return MVP_EBADSHAPE;
  if (b == 0 || cw == 0 || n == 0) return MVP_OK;
  if (!out || !v || (k > 0 && (!w || !idx))) return MVP_EBADARG;
  const int chunk = gs_chunk(b, cw, n);
  const dim3 grid((n + chunk - 1) / chunk, cw, b);
  hipStream_t st = as_stream(stream);
#define MVP_GS(S) hipLaunchKernelGGL((share_gather_sum_kernel<S, false>), grid, dim3(kGsThreads), 0, st, cw, k, n_src, n, chunk, w, v, idx, nullptr, out, nullptr, nullptr)
  switch (share) {
    case 1: MVP_GS(1); break;
    case 2: MVP_GS(2); break;
    case 4: MVP_GS(4); break;
    case 8: MVP_GS(8); break;
    default: MVP_GS(16); break;
  }
#undef MVP_GS
  return check_launch("mvp_share_gather_sum");
}

extern "C" int mvp_share_gather_sum_grad(int b, int share, int cw, int k, int n_src, int n, const float *w, const float *v,
                                         const int *idx, const float *grad_out, float *grad_w, float *grad_vals,
                                         void *stream) {
  if (!ag_shape_ok(b, share, cw, k, n) || n_src <= 0 || (long long)share * n_src * 4 > kGsMaxLds) return MVP_EBADSHAPE;
  if (b == 0 || cw == 0 || n == 0 || k == 0) return MVP_OK;
  if (!w || !v || !idx || !grad_out || !grad_w || !grad_vals) return MVP_EBADARG;
  const int chunk = gs_chunk(b, cw, n);
  const dim3 grid((n + chunk - 1) / chunk, cw, b);
  hipStream_t st = as_stream(stream);
#define MVP_GS(S) hipLaunchKernelGGL((share_gather_sum_kernel<S, true>), grid, dim3(kGsThreads), 0, st, cw, k, n_src, n, chunk, w, v, idx, grad_out, nullptr, grad_w, grad_vals)
  switch (share) {
    case 1: MVP_GS(1); break;
    case 2: MVP_GS(2); break;
    case 4: MVP_GS(4); break;
    case 8: MVP_GS(8); break;
    default: MVP_GS(16); break;
  }
#undef MVP_GS
  return check_launch("mvp_share_gather_sum_grad");
}

extern "C" int mvp_share_weighted_sum_grad(int b, int share, int cw, int k, int n, const float *w, const float *v,
                                           const float *grad_out, float *grad_w, float *grad_v, void *stream) {
  if (!ag_shape_ok(b, share, cw, k, n)) return MVP_EBADSHAPE;
  if (b == 0 || cw == 0 || n == 0 || k == 0) return MVP_OK;
  if (!w || !v || !grad_out || !grad_w || !grad_v) return MVP_EBADARG;
  const dim3 grid((n + kAgThreads - 1) / kAgThreads, cw, b);
  hipStream_t st = as_stream(stream);
  MVP_AG_DISPATCH(share_weighted_sum_grad_kernel, cw, k, n, w, v, grad_out, grad_w, grad_v)
  return check_launch("mvp_share_weighted_sum_grad");
}
