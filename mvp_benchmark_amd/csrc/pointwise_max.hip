// Backward pass of "1x1 convolution followed by a max over the positions" -- the PointNet stage
// of the completion networks (completion/models/pcn.py:25-31 `conv4(x)` then `torch.max(x, 2)`;
// vrcnet.py conv5 of the relational encoder, ecg.py gf_conv): y = W x + bias on x (b, cin, len),
// v[b][co] = max_l y[b][co][l].  Under autograd the max hands the convolution a (b, cout, len)
// gradient that is zero except at the b * cout winning positions, and the library (or a dense GEMM
// kernel) then runs its data- and weight-gradient passes over all b * len positions of that tensor
// of zeros: 2 * 2 * b * cin * cout * len flops (1.16 + 1.40 ms at 64 x (512 -> 1024) x 2048 on the
// MFMA kernels of pointwise_mfma.hip).  Here both gradients walk the winners only:
//   gw[co][ci]   = sum_b g[b][co] x[b][ci][idx[b][co]]                (a gather of b columns of x)
//   gx[b][ci][l] = sum_{co : idx[b][co] = l} g[b][co] W[co][ci]       (0 for positions nobody won)
// with g (b, cout) the gradient of v and idx (b, cout) the winning positions; 2 * b * cin * cout
// multiply-adds each.  Fixed summation orders (b ascending; co ascending): bit-reproducible.
#include "common.h"

namespace mvp {

constexpr int kMxThreads = 256;

// gw, gb: one workgroup per output channel; a thread owns input channels ci = t, t + 256, ... and
// walks the clouds in order (independent gathers, eight in flight).
__global__ __launch_bounds__(kMxThreads) void convmax_wgrad_kernel(int b, int cin, int cout, int len,
                                                                  const float *__restrict__ x,
                                                                  const float *__restrict__ g,
                                                                  const int *__restrict__ idx,
                                                                  float *__restrict__ gw, float *__restrict__ gb) {
  const int co = blockIdx.x, t = threadIdx.x;
  for (int ci = t; ci < cin; ci += kMxThreads) {
    float s = 0.f;
    int c = 0;
    for (; c + 8 <= b; c += 8) {
      float v[8], w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int l = min(max(idx[(size_t)(c + u) * cout + co], 0), len - 1);
        w[u] = g[(size_t)(c + u) * cout + co];
        v[u] = x[((size_t)(c + u) * cin + ci) * len + l];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s = __builtin_fmaf(w[u], v[u], s);
    }
    for (; c < b; ++c)
      s = __builtin_fmaf(g[(size_t)c * cout + co], x[((size_t)c * cin + ci) * len + min(max(idx[(size_t)c * cout + co], 0), len - 1)], s);
    gw[(size_t)co * cin + ci] = s;
  }
  if (gb && t == 0) {
    float s = 0.f;
    for (int c = 0; c < b; ++c) s += g[(size_t)c * cout + co];
    gb[co] = s;
  }
}

// gx: a workgroup (4 waves) owns one cloud and 64 input channels.  It groups the output channels by the
// position they won (LDS counting sort: atomic counts, prefix sum over the positions, atomic fill, then every
// thread sorts the short groups of its own positions by channel, so that the sums below have a fixed order).
// A wave then takes tiles of 32 positions: the channels of a tile are a contiguous run of the sorted list; lane =
// input channel reads their weight rows coalesced (eight rows in flight) and accumulates g W into a 32 x 64 LDS
// tile, which is written out transposed -- lanes along the positions, 128-byte segments -- so that EVERY entry of
// gx is stored exactly once, coalesced, zeros included.
constexpr int kMxMaxLen = 16384, kMxMaxCout = 4096, kMxBlock = 256, kMxTL = 32, kMxTC = 64, kMxTStride = kMxTC + 1;

__global__ __launch_bounds__(kMxBlock) void convmax_dgrad_kernel(int cin, int cout, int len,
                                                                const float *__restrict__ w,
                                                                const float *__restrict__ g,
                                                                const int *__restrict__ idx,
                                                                float *__restrict__ gx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int *s_idx = reinterpret_cast<int *>(smem);                 // cout
  float *s_g = reinterpret_cast<float *>(s_idx + cout);       // cout
  int *s_list = reinterpret_cast<int *>(s_g + cout);          // cout: channels grouped by position
  int *s_end = s_list + cout;                                 // len: counts, then cursors, then END of a position's group
  __shared__ float s_tile[kMxBlock / 64][kMxTL * kMxTStride];
  __shared__ int s_wave[kMxBlock / 64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int cloud = blockIdx.x, ci0 = blockIdx.y * kMxTC;
  for (int c = t; c < cout; c += kMxBlock) {
    s_idx[c] = min(max(idx[(size_t)cloud * cout + c], 0), len - 1);
    s_g[c] = g[(size_t)cloud * cout + c];
  }
  for (int l = t; l < len; l += kMxBlock) s_end[l] = 0;
  __syncthreads();
  for (int c = t; c < cout; c += kMxBlock) atomicAdd(&s_end[s_idx[c]], 1);
  __syncthreads();
  // exclusive prefix sum over the positions (a thread owns a contiguous run of positions)
  const int per = (len + kMxBlock - 1) / kMxBlock;
  const int l_begin = min(len, t * per), l_end = min(len, l_begin + per);
  int mine = 0;
  for (int l = l_begin; l < l_end; ++l) mine += s_end[l];
  int scan = mine;                                            // inclusive wave scan
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(scan, d, 64);
    if (lane >= d) scan += o;
  }
  if (lane == 63) s_wave[wave] = scan;
  __syncthreads();
  int first = scan - mine;                                    // start of this thread's first group
  for (int wv = 0; wv < wave; ++wv) first += s_wave[wv];
  {
    int off = first;
    for (int l = l_begin; l < l_end; ++l) {
      const int c = s_end[l];
      s_end[l] = off;                                         // cursor = start
      off += c;
    }
  }
  __syncthreads();
  for (int c = t; c < cout; c += kMxBlock) s_list[atomicAdd(&s_end[s_idx[c]], 1)] = c;   // cursor ends at the group's end
  __syncthreads();
  {                                                           // channels ascending within a group
    int a = first;
    for (int l = l_begin; l < l_end; ++l) {
      const int e = s_end[l];
      for (int q = a + 1; q < e; ++q) {
        const int v = s_list[q];
        int r = q - 1;
        while (r >= a && s_list[r] > v) {
          s_list[r + 1] = s_list[r];
          --r;
        }
        s_list[r + 1] = v;
      }
      a = e;
    }
  }
  __syncthreads();

  float *tile = s_tile[wave];
  float *out = gx + (size_t)cloud * cin * len;
  const int ci = ci0 + lane;
  const bool ci_ok = ci < cin;
  const int hl = lane & 31, hc = lane >> 5;                   // write-out: 32 positions x 2 channels per store
  for (int l0 = wave * kMxTL; l0 < len; l0 += (kMxBlock / 64) * kMxTL) {
    const int l1 = min(len, l0 + kMxTL);
    const int a = l0 ? s_end[l0 - 1] : 0, e = s_end[l1 - 1];  // the tile's run of the sorted list (wave-uniform)
    if (a == e) {                                             // nobody won here: zeros straight to memory
      if (l0 + hl < l1)
        for (int r = 0; r < kMxTC; r += 2)
          if (ci0 + r + hc < cin) out[(size_t)(ci0 + r + hc) * len + l0 + hl] = 0.f;
      continue;
    }
#pragma unroll
    for (int r = 0; r < kMxTL; ++r) tile[r * kMxTStride + lane] = 0.f;
    for (int q = a; q < e; q += 8) {
      float wv[8], gv[8];
      int row[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = s_list[min(q + u, e - 1)];
        row[u] = (s_idx[c] - l0) * kMxTStride + lane;
        gv[u] = q + u < e ? s_g[c] : 0.f;                     // (the run's tail: the last entry again, times 0)
        wv[u] = ci_ok ? w[(size_t)c * cin + ci] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (q + u < e) tile[row[u]] = __builtin_fmaf(gv[u], wv[u], tile[row[u]]);
    }
    if (l0 + hl < l1)
      for (int r = 0; r < kMxTC; r += 2)
        if (ci0 + r + hc < cin) out[(size_t)(ci0 + r + hc) * len + l0 + hl] = tile[hl * kMxTStride + r + hc];
  }
}

}  // namespace mvp

using namespace mvp;

extern "C" int mvp_pointwise_max_backward(int b, int cin, int cout, int len, const float *x, const float *w,
                                          const float *g, const int *idx, float *gx, float *gw, float *gb,
                                          void *stream) {
  if (b <= 0 || cin <= 0 || cout <= 0 || len <= 0) return MVP_EBADSHAPE;
  if (len > kMxMaxLen || cout > kMxMaxCout || b > 65535) return MVP_EBADSHAPE;
  if (gx && (size_t)(3 * cout + len) * 4 > 30000) return MVP_EBADSHAPE;   // the grouping's LDS (+ 33 KB of tiles)
  if (!g || !idx || (gw && !x) || (gx && !w)) return MVP_EBADARG;
  hipStream_t st = as_stream(stream);
  if (gw)
    hipLaunchKernelGGL(convmax_wgrad_kernel, dim3(cout), dim3(kMxThreads), 0, st, b, cin, cout, len, x, g, idx, gw, gb);
  if (gx) {
    const size_t lds = (size_t)(3 * cout + len) * 4;
    hipLaunchKernelGGL(convmax_dgrad_kernel, dim3(b, (cin + kMxTC - 1) / kMxTC), dim3(kMxBlock), lds, st, cin, cout, len,
                       w, g, idx, gx);
  }
  return check_launch("mvp_pointwise_max_backward");
}
