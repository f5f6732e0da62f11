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
constexpr int kMxMaxLen = 16384, kMxMaxCout = 4096, kMxBlock = 256, kMxTL = 32, kMxTC = 64, kMxTStride = kMxTC + 1;

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

// gx: a workgroup (4 waves) owns one cloud and 64 input channels.  It sorts the output channels by the position
// they won -- keys {position, channel} through a bitonic network in LDS: deterministic, and indifferent to how
// the winners are distributed (a trained PointNet concentrates hundreds of channels on a few critical points) --
// so that a position's channels are a contiguous, ascending run of the list.
// A wave then takes tiles of 32 positions: the channels of a tile are a contiguous run of the sorted list; lane =
// input channel reads their weight rows coalesced (eight rows in flight) and accumulates g W into a 32 x 64 LDS
// tile, which is written out transposed -- lanes along the positions, 128-byte segments -- so that EVERY entry of
// gx is stored exactly once, coalesced, zeros included.

// Sorts one cloud's output channels by the position they won (keys position << 12 | channel, bitonic network in
// LDS) and leaves in s_end[l] the END of position l's run of the sorted list (a run starts where the previous
// position's ends).  All threads of the workgroup (kMxBlock) call it; s_key holds cout_p2 = 2^k >= cout entries.
__device__ __forceinline__ void convmax_sort_runs(int cout, int cout_p2, int len, int cloud, const float *__restrict__ g,
                                                  const int *__restrict__ idx, int *s_key, float *s_g, int *s_end,
                                                  int *s_wave) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int c = t; c < cout_p2; c += kMxBlock) {
    if (c < cout) {
      s_key[c] = (min(max(idx[(size_t)cloud * cout + c], 0), len - 1) << 12) | c;
      if (s_g) s_g[c] = g[(size_t)cloud * cout + c];
    } else {
      s_key[c] = 0x7FFFFFFF;                                  // padding sorts to the end
    }
  }
  for (int l = t; l < len; l += kMxBlock) s_end[l] = 0;
  __syncthreads();
  for (int k = 2; k <= cout_p2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < cout_p2; i += kMxBlock) {
        const int o = i ^ j;
        if (o > i) {
          const int a = s_key[i], b = s_key[o];
          if ((a > b) == ((i & k) == 0)) {
            s_key[i] = b;
            s_key[o] = a;
          }
        }
      }
      __syncthreads();
    }
  // a run's end at the position of its last entry, then a running maximum over the positions fills in the
  // positions nobody won (their run is empty: it ends where the previous one does)
  for (int i = t; i < cout; i += kMxBlock) {
    const int p = s_key[i] >> 12;
    if (i + 1 == cout || (s_key[i + 1] >> 12) != p) s_end[p] = i + 1;
  }
  __syncthreads();
  {
    const int per = (len + kMxBlock - 1) / kMxBlock;
    const int l_begin = min(len, t * per), l_end = min(len, l_begin + per);
    int mine = 0;
    for (int l = l_begin; l < l_end; ++l) mine = max(mine, s_end[l]);
    int scan = mine;                                          // inclusive wave scan (max)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(scan, d, 64);
      if (lane >= d) scan = max(scan, o);
    }
    if (lane == 63) s_wave[wave] = scan;
    __syncthreads();
    int run = __shfl_up(scan, 1, 64);                         // maximum over the threads before this one
    if (lane == 0) run = 0;
    for (int wv = 0; wv < wave; ++wv) run = max(run, s_wave[wv]);
    for (int l = l_begin; l < l_end; ++l) {
      run = max(run, s_end[l]);
      s_end[l] = run;
    }
  }
  __syncthreads();

}

__global__ __launch_bounds__(kMxBlock) void convmax_dgrad_kernel(int cin, int cout, int cout_p2, int len,
                                                                const float *__restrict__ w,
                                                                const float *__restrict__ g,
                                                                const int *__restrict__ idx,
                                                                float *__restrict__ gx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int *s_key = reinterpret_cast<int *>(smem);                 // cout_p2: position << 12 | channel, sorted
  float *s_g = reinterpret_cast<float *>(s_key + cout_p2);    // cout
  int *s_end = reinterpret_cast<int *>(s_g + cout);           // len: END of a position's run of the sorted list
  __shared__ float s_tile[kMxBlock / 64][kMxTL * kMxTStride];
  __shared__ int s_wave[kMxBlock / 64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int cloud = blockIdx.x, ci0 = blockIdx.y * kMxTC;
  convmax_sort_runs(cout, cout_p2, len, cloud, g, idx, s_key, s_g, s_end, s_wave);

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
        const int key = s_key[min(q + u, e - 1)];
        const int c = key & 4095;
        row[u] = ((key >> 12) - l0) * kMxTStride + lane;
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

// Weight gradient, coalesced: the workgroup of (cloud, 64 input channels) sorts as above; a wave stages the tiles
// of 32 positions that somebody won through LDS (reads along the positions, 128-byte segments) and writes, for
// every channel c of the tile's run, the column it won -- cols[b][c][ci] = x[b][ci][idx[b][c]] -- as a 256-byte row;
// convmax_wgrad_reduce_kernel then adds g[b][c] cols[b][c][:] over the clouds in order.  (The direct gather of
// convmax_wgrad_kernel touches a 64-byte sector per 4-byte value: 0.54 ms at 64 x (512 -> 1024) x 2048.)
__global__ __launch_bounds__(kMxBlock) void convmax_cols_kernel(int cin, int cout, int cout_p2, int len,
                                                               const float *__restrict__ x, const int *__restrict__ idx,
                                                               float *__restrict__ cols) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int *s_key = reinterpret_cast<int *>(smem);
  int *s_end = s_key + cout_p2;
  __shared__ float s_tile[kMxBlock / 64][kMxTL * kMxTStride];
  __shared__ int s_wave[kMxBlock / 64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int cloud = blockIdx.x, ci0 = blockIdx.y * kMxTC;
  convmax_sort_runs(cout, cout_p2, len, cloud, nullptr, idx, s_key, nullptr, s_end, s_wave);
  // The four waves share a tile of 64 positions x 64 channels (the whole s_tile area): a trained network's winners
  // sit on a few critical points, and the run of one tile can hold hundreds of channels.  A wave reads its 16 rows
  // with the lanes along the positions (256-byte segments); the NEXT non-empty tile's rows are already in flight
  // while this tile's columns are written out.
  constexpr int kTL = 64, kRows = kMxTC / (kMxBlock / 64);      // positions per tile; rows per wave
  static_assert((kMxBlock / 64) * kMxTL * kMxTStride >= kTL * kMxTStride, "the shared tile fits the per-wave tiles' area");
  float *tile = &s_tile[0][0];
  const float *in = x + (size_t)cloud * cin * len;
  const bool ci_ok = ci0 + lane < cin;
  auto run_of = [&](int l0, int &a, int &e) {                  // the tile's run of the sorted list (uniform over the workgroup)
    a = l0 ? s_end[l0 - 1] : 0;
    e = s_end[min(len, l0 + kTL) - 1];
  };
  auto next_tile = [&](int l0) {                               // first non-empty tile at or after l0 (len: none)
    for (; l0 < len; l0 += kTL) {
      int a, e;
      run_of(l0, a, e);
      if (a != e) return l0;
    }
    return len;
  };
  float v[kRows];
  auto fetch = [&](int l0) {
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
      const int ci = ci0 + wave * kRows + r;
      v[r] = (l0 + lane < len && ci < cin) ? in[(size_t)ci * len + l0 + lane] : 0.f;
    }
  };
  int l0 = next_tile(0);
  if (l0 < len) fetch(l0);
  while (l0 < len) {
    __syncthreads();                                           // the previous tile has been read
#pragma unroll
    for (int r = 0; r < kRows; ++r) tile[lane * kMxTStride + wave * kRows + r] = v[r];
    const int ln = next_tile(l0 + kTL);
    if (ln < len) fetch(ln);                                   // in flight while this tile's columns go out
    __syncthreads();
    int a, e;
    run_of(l0, a, e);
    for (int q = a + wave; q < e; q += kMxBlock / 64) {
      const int key = s_key[q];
      if (ci_ok) cols[((size_t)cloud * cout + (key & 4095)) * cin + ci0 + lane] = tile[((key >> 12) - l0) * kMxTStride + lane];
    }
    l0 = ln;
  }
}

__global__ __launch_bounds__(kMxThreads) void convmax_wgrad_reduce_kernel(int b, int cin, int cout,
                                                                         const float *__restrict__ cols,
                                                                         const float *__restrict__ g,
                                                                         float *__restrict__ gw, float *__restrict__ gb) {
  const int co = blockIdx.x, t = threadIdx.x;
  for (int ci = t; ci < cin; ci += kMxThreads) {
    float s = 0.f;
    int c = 0;
    for (; c + 8 <= b; c += 8) {                               // eight independent loads in flight, the sum in cloud order
      float v[8], w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        w[u] = g[(size_t)(c + u) * cout + co];
        v[u] = cols[((size_t)(c + u) * cout + co) * cin + ci];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s = __builtin_fmaf(w[u], v[u], s);
    }
    for (; c < b; ++c) s = __builtin_fmaf(g[(size_t)c * cout + co], cols[((size_t)c * cout + co) * cin + ci], s);
    gw[(size_t)co * cin + ci] = s;
  }
  if (gb && t == 0) {
    float s = 0.f;
    for (int c = 0; c < b; ++c) s += g[(size_t)c * cout + co];
    gb[co] = s;
  }
}

}  // namespace mvp

using namespace mvp;

extern "C" long long mvp_pointwise_max_backward_scratch_bytes(int b, int cin, int cout, int len) {
  if (b <= 0 || cin <= 0 || cout <= 0 || len <= 0 || len > kMxMaxLen || cout > kMxMaxCout) return 0;
  if ((size_t)(3 * cout + len) * 4 > 30000) return 0;
  return (long long)b * cout * cin * 4;
}

extern "C" int mvp_pointwise_max_backward(int b, int cin, int cout, int len, const float *x, const float *w,
                                          const float *g, const int *idx, float *gx, float *gw, float *gb,
                                          void *scratch, long long scratch_bytes, void *stream) {
  if (b <= 0 || cin <= 0 || cout <= 0 || len <= 0) return MVP_EBADSHAPE;
  if (len > kMxMaxLen || cout > kMxMaxCout || b > 65535) return MVP_EBADSHAPE;
  if (gx && (size_t)(3 * cout + len) * 4 > 30000) return MVP_EBADSHAPE;   // the sort's LDS (+ 33 KB of tiles)
  if (!g || !idx || (gw && !x) || (gx && !w)) return MVP_EBADARG;
  hipStream_t st = as_stream(stream);
  int cout_p2 = 1;
  while (cout_p2 < cout) cout_p2 <<= 1;
  const dim3 grid(b, (cin + kMxTC - 1) / kMxTC);
  if (gw) {
    const long long need = mvp_pointwise_max_backward_scratch_bytes(b, cin, cout, len);
    if (scratch && need > 0 && scratch_bytes >= need) {       // staged columns: coalesced
      float *cols = static_cast<float *>(scratch);
      hipLaunchKernelGGL(convmax_cols_kernel, grid, dim3(kMxBlock), (size_t)(cout_p2 + len) * 4, st, cin, cout, cout_p2,
                         len, x, idx, cols);
      hipLaunchKernelGGL(convmax_wgrad_reduce_kernel, dim3(cout), dim3(kMxThreads), 0, st, b, cin, cout, cols, g, gw, gb);
    } else {
      hipLaunchKernelGGL(convmax_wgrad_kernel, dim3(cout), dim3(kMxThreads), 0, st, b, cin, cout, len, x, g, idx, gw, gb);
    }
  }
  if (gx) {
    const size_t lds = (size_t)(cout_p2 + cout + len) * 4;
    hipLaunchKernelGGL(convmax_dgrad_kernel, grid, dim3(kMxBlock), lds, st, cin, cout, cout_p2, len, w, g, idx, gx);
  }
  return check_launch("mvp_pointwise_max_backward");
}
