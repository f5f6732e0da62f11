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
