// Per-point / per-edge linear maps (1x1 convolutions) on the matrix cores.
//
// the grouped-feature MLPs of the completion networks
// (completion/model_utils.py:26-55 EF_expansion, completion/models/vrcnet.py:21-57
// SA_module, ecg.py:36-65 Dense_conv, pcn.py encoder / decoder convolutions) are
// 1x1 convolutions, i.e. per cloud  Y (Cout x L) = W (Cout x Cin) X (Cin x L) with L
// = points (x neighbours) contiguous.  The reference runs them as cuDNN
// convolutions followed by separate bias / ReLU / residual / max-over-neighbours
// kernels.  Here: one LDS-tiled GEMM on v_mfma_f32_32x32x2_f32 -- float32 in,
// float32 accumulate, bit-for-bit a k-ordered fmaf chain (MI355X_MICROARCH.md:
// gfx950 has no xf32/TF32; the f32 MFMA peak is 157 TFLOP/s, the same as the
// packed vector peak, but reachable from one wave per SIMD) -- with the epilogue
// fused: + bias, ReLU, + residual, and max over groups of `group` consecutive
// columns (the neighbours of one point), so that the (B, C, P, S) tensor of a
// set-abstraction stage is never written.
//
// Tiling (256 threads = 4 waves in a 2 x 2 grid): block tile BM x 128 columns,
// BM = 128 (wave: 2 x 2 MFMA blocks of 32 x 32, 64 accumulator VGPRs) or 64
// (wave: 1 x 2), K in slabs of 16 through double-buffered LDS (33 KiB: four
// workgroups per CU hide the global latency; the slab for step i+1 is fetched
// into registers while step i computes).  A fragments are read from an
// [k][m]-major image (lanes = consecutive rows), B fragments from [k][n]
// (lanes = consecutive columns): conflict-free ds_read_b32, one per MFMA operand
// -- the f32 MFMA issues every 64 cycles, LDS bandwidth is not a concern.
// The weight may be given (Cout, Cin) row-major (forward) or as its transpose
// (the data gradient multiplies by W^T: pass the forward weight and w_kmajor = 1).
#include "common.h"

namespace mvp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef MVP_MM_BK
#define MVP_MM_BK 16
#endif
constexpr int kMmBN = 128, kMmBK = MVP_MM_BK, kMmThreads = 256;

template <int BM, bool KMAJOR>
__global__ __launch_bounds__(kMmThreads) void pointwise_mfma_kernel(
    int cin, int cout, int len, const float *__restrict__ x, const float *__restrict__ xmask,
    const float *__restrict__ w, const float *__restrict__ bias, const float *__restrict__ residual, int relu,
    int group, float *__restrict__ y) {
  constexpr int TM = BM / 64;        // MFMA blocks per wave along M
  constexpr int TN = 2;              // ... along N (wave tile: 32 TM x 64)
  constexpr int LDA = BM + 2;        // [k][m] image, padded: rows k and k + 1 (lanes 32..63) land on shifted banks
  __shared__ __attribute__((aligned(16))) float As[2][kMmBK][LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][kMmBK][kMmBN];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.x * kMmBN, m0 = blockIdx.y * BM, cloud = blockIdx.z;
  const float *xb = x + (size_t)cloud * cin * len;
  const float *mb = xmask ? xmask + (size_t)cloud * cin * len : nullptr;   // x is used where mask > 0 (ReLU'(.))

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- global -> registers (one K slab): A BM x 16, B 16 x 128
  constexpr int A4 = BM * kMmBK / 4 / kMmThreads;   // float4 per thread: 2 (BM 128) or 1 (BM 64)
  constexpr int B4f = kMmBK * kMmBN / 4 / kMmThreads;  // float4 of the B slab per thread
  float4 ra[A4], rb[B4f];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A4; ++i) {
      const int q = t + i * kMmThreads;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (KMAJOR) {            // w is (cin, cout): 4 consecutive m of one k
        const int k = q / (BM / 4), m = (q % (BM / 4)) * 4;
        if (k0 + k < cin) {
          const float *p = w + (size_t)(k0 + k) * cout + m0 + m;
          if (m0 + m + 3 < cout) {
            v = *reinterpret_cast<const float4 *>(p);
          } else {
            if (m0 + m < cout) v.x = p[0];
            if (m0 + m + 1 < cout) v.y = p[1];
            if (m0 + m + 2 < cout) v.z = p[2];
          }
        }
      } else {                            // w is (cout, cin): 4 consecutive k of one m
        const int m = q / (kMmBK / 4), k = (q % (kMmBK / 4)) * 4;
        if (m0 + m < cout) {
          const float *p = w + (size_t)(m0 + m) * cin + k0 + k;
          if (k0 + k + 3 < cin && (cin & 3) == 0) {
            v = *reinterpret_cast<const float4 *>(p);
          } else {
            if (k0 + k < cin) v.x = p[0];
            if (k0 + k + 1 < cin) v.y = p[1];
            if (k0 + k + 2 < cin) v.z = p[2];
            if (k0 + k + 3 < cin) v.w = p[3];
          }
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B4f; ++i) {
      const int q = t + i * kMmThreads;
      const int k = q / 32, n = (q % 32) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k0 + k < cin && n0 + n < len) {   // len % 4 == 0: a float4 is inside or outside
        v = *reinterpret_cast<const float4 *>(xb + (size_t)(k0 + k) * len + n0 + n);
        if (mb) {
          const float4 mk = *reinterpret_cast<const float4 *>(mb + (size_t)(k0 + k) * len + n0 + n);
          v.x = mk.x > 0.f ? v.x : 0.f;
          v.y = mk.y > 0.f ? v.y : 0.f;
          v.z = mk.z > 0.f ? v.z : 0.f;
          v.w = mk.w > 0.f ? v.w : 0.f;
        }
      }
      rb[i] = v;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A4; ++i) {
      const int q = t + i * kMmThreads;
      if constexpr (KMAJOR) {
        const int k = q / (BM / 4), m = (q % (BM / 4)) * 4;
        As[buf][k][m] = ra[i].x;
        As[buf][k][m + 1] = ra[i].y;
        As[buf][k][m + 2] = ra[i].z;
        As[buf][k][m + 3] = ra[i].w;
      } else {
        const int m = q / (kMmBK / 4), k = (q % (kMmBK / 4)) * 4;
        As[buf][k][m] = ra[i].x;
        As[buf][k + 1][m] = ra[i].y;
        As[buf][k + 2][m] = ra[i].z;
        As[buf][k + 3][m] = ra[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < B4f; ++i) {
      const int q = t + i * kMmThreads;
      *reinterpret_cast<float4 *>(&Bs[buf][q / 32][(q % 32) * 4]) = rb[i];
    }
  };

  const int nk = (cin + kMmBK - 1) / kMmBK;
  fetch(0);
  stash(0);
  __syncthreads();
  const int lrow = lane & 31, lk = lane >> 5;
  for (int s = 0; s < nk; ++s) {
    const int buf = s & 1;
    if (s + 1 < nk) fetch((s + 1) * kMmBK);      // in flight while this slab computes
#pragma unroll
    for (int kk = 0; kk < kMmBK; kk += 2) {
      float a[TM], bq[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[buf][kk + lk][wm * 32 * TM + i * 32 + lrow];
#pragma unroll
      for (int j = 0; j < TN; ++j) bq[j] = Bs[buf][kk + lk][wn * 64 + j * 32 + lrow];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bq[j], acc[i][j], 0, 0, 0);
    }
    if (s + 1 < nk) {
      stash(buf ^ 1);                              // the other buffer: nobody reads it in this step
      __syncthreads();
    }
  }

  // ---- epilogue: C/D layout of 32x32: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const int len_out = len / group;
  float *yb = y + (size_t)cloud * cout * len_out;
  const float *rbse = residual ? residual + (size_t)cloud * cout * len_out : nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * 64 + j * 32 + lrow;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        float v = acc[i][j][r];
        if (bias && row < cout) v += bias[row];
        if (relu) v = __builtin_fmaxf(v, 0.f);
        if (group > 1) {
          if (col >= len) v = -__builtin_inff();     // (len % group == 0: a group is inside or outside)
          for (int off = 1; off < group; off <<= 1) v = __builtin_fmaxf(v, __shfl_xor(v, off, 64));
        }
        if (row < cout && col < len && (lrow & (group - 1)) == 0) {
          const size_t o = (size_t)row * len_out + col / group;
          if (rbse) v += rbse[o];
          yb[o] = v;
        }
      }
    }
  }
}


// ---------------------------------------------------------------- weight gradient
// gw[co][ci] = sum_b sum_l g[b][co][l] x[b][ci][l]   (g = grad_out, optionally masked by ReLU'),
// gb[co] = sum_b sum_l g[b][co][l]: the same GEMM with one more "input channel" that is all ones.
// M = cout, N = cin (+1), K = the b * len positions -- a huge reduction with a small output:
// the positions are split over `splits` workgroups per output tile, each writes its partial tile,
// a second kernel adds the partials in a fixed order (no float atomics: reproducible).
// Both operands are K-contiguous in memory, so both LDS images are [row][k] (padded to an odd
// stride: a fragment read takes 32 rows at one k).
constexpr int kWgBK = 32;

template <int BM>
__global__ __launch_bounds__(kMmThreads) void pointwise_wgrad_mfma_kernel(
    int b, int cin, int cout, int len, int chunk, int chunks_per_cloud, const float *__restrict__ x,
    const float *__restrict__ g, const float *__restrict__ gmask, int with_bias, float *__restrict__ partial) {
  constexpr int TM = BM / 64, TN = 2;
  constexpr int LD = kWgBK + 1;
  __shared__ float As[2][BM][LD];
  __shared__ float Bs[2][kMmBN][LD];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ncols = cin + (with_bias ? 1 : 0);
  const int tiles_n = (ncols + kMmBN - 1) / kMmBN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * kMmBN;
  const int split = blockIdx.y;                     // one (cloud, chunk of positions) per split
  const int cloud = split / chunks_per_cloud, l0 = (split % chunks_per_cloud) * chunk;
  const int l1 = min(len, l0 + chunk);
  const float *gb_ = g + (size_t)cloud * cout * len;
  const float *mb = gmask ? gmask + (size_t)cloud * cout * len : nullptr;
  const float *xb = x + (size_t)cloud * cin * len;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  constexpr int A4 = BM * kWgBK / 4 / kMmThreads;    // float4 per thread: 4 (BM 128) or 2 (BM 64)
  constexpr int B4 = kMmBN * kWgBK / 4 / kMmThreads; // 4
  float4 ra[A4], rb[B4];
  auto fetch = [&](int k0) {                         // k0: first position of the slab
#pragma unroll
    for (int i = 0; i < A4; ++i) {
      const int q = t + i * kMmThreads;
      const int m = q / (kWgBK / 4), k = (q % (kWgBK / 4)) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m0 + m < cout && k0 + k < l1) {            // len % 4 == 0 and chunk % 4 == 0
        v = *reinterpret_cast<const float4 *>(gb_ + (size_t)(m0 + m) * len + k0 + k);
        if (mb) {
          const float4 mk = *reinterpret_cast<const float4 *>(mb + (size_t)(m0 + m) * len + k0 + k);
          v.x = mk.x > 0.f ? v.x : 0.f;
          v.y = mk.y > 0.f ? v.y : 0.f;
          v.z = mk.z > 0.f ? v.z : 0.f;
          v.w = mk.w > 0.f ? v.w : 0.f;
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B4; ++i) {
      const int q = t + i * kMmThreads;
      const int j = q / (kWgBK / 4), k = (q % (kWgBK / 4)) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k0 + k < l1) {
        if (n0 + j < cin) v = *reinterpret_cast<const float4 *>(xb + (size_t)(n0 + j) * len + k0 + k);
        else if (n0 + j == cin && with_bias) v = make_float4(1.f, 1.f, 1.f, 1.f);   // the bias column
      }
      rb[i] = v;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A4; ++i) {
      const int q = t + i * kMmThreads;
      const int m = q / (kWgBK / 4), k = (q % (kWgBK / 4)) * 4;
      As[buf][m][k] = ra[i].x;
      As[buf][m][k + 1] = ra[i].y;
      As[buf][m][k + 2] = ra[i].z;
      As[buf][m][k + 3] = ra[i].w;
    }
#pragma unroll
    for (int i = 0; i < B4; ++i) {
      const int q = t + i * kMmThreads;
      const int j = q / (kWgBK / 4), k = (q % (kWgBK / 4)) * 4;
      Bs[buf][j][k] = rb[i].x;
      Bs[buf][j][k + 1] = rb[i].y;
      Bs[buf][j][k + 2] = rb[i].z;
      Bs[buf][j][k + 3] = rb[i].w;
    }
  };
  const int nk = (l1 - l0 + kWgBK - 1) / kWgBK;
  const int lrow = lane & 31, lk = lane >> 5;
  if (nk > 0) {
    fetch(l0);
    stash(0);
  }
  __syncthreads();
  for (int s = 0; s < nk; ++s) {
    const int buf = s & 1;
    if (s + 1 < nk) fetch(l0 + (s + 1) * kWgBK);
#pragma unroll
    for (int kk = 0; kk < kWgBK; kk += 2) {
      float a[TM], bq[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[buf][wm * 32 * TM + i * 32 + lrow][kk + lk];
#pragma unroll
      for (int j = 0; j < TN; ++j) bq[j] = Bs[buf][wn * 64 + j * 32 + lrow][kk + lk];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bq[j], acc[i][j], 0, 0, 0);
    }
    if (s + 1 < nk) {
      stash(buf ^ 1);
      __syncthreads();
    }
  }
  // partial[split][co][col], col < ncols
  float *pp = partial + (size_t)split * cout * ncols;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * 64 + j * 32 + lrow;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row < cout && col < ncols) pp[(size_t)row * ncols + col] = acc[i][j][r];
      }
    }
}

// gw[co][ci] (and gb[co]) = sum over the splits, in split order
__global__ __launch_bounds__(256) void pointwise_wgrad_reduce_kernel(int cin, int cout, int with_bias, int splits,
                                                                     const float *__restrict__ partial,
                                                                     float *__restrict__ gw, float *__restrict__ gb) {
  const int ncols = cin + (with_bias ? 1 : 0);
  const long long total = (long long)cout * ncols;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int sp = 0;
  for (; sp + 3 < splits; sp += 4) {
    s0 += partial[(size_t)sp * total + e];
    s1 += partial[(size_t)(sp + 1) * total + e];
    s2 += partial[(size_t)(sp + 2) * total + e];
    s3 += partial[(size_t)(sp + 3) * total + e];
  }
  for (; sp < splits; ++sp) s0 += partial[(size_t)sp * total + e];
  const float v = (s0 + s1) + (s2 + s3);
  const int row = (int)(e / ncols), col = (int)(e % ncols);
  if (col < cin) gw[(size_t)row * cin + col] = v;
  else if (gb) gb[row] = v;
}

// positions per split: whole clouds are cut into chunks so that ~2048 workgroups exist
static void wgrad_plan(int b, int cin, int cout, int len, int with_bias, int &chunk, int &chunks_per_cloud, int &tiles) {
  const int bm = cout > 64 ? 128 : 64;
  const int ncols = cin + (with_bias ? 1 : 0);
  tiles = ((cout + bm - 1) / bm) * ((ncols + kMmBN - 1) / kMmBN);
  int want = (2048 + tiles - 1) / tiles;             // splits wanted
  chunks_per_cloud = (want + b - 1) / b;
  if (chunks_per_cloud < 1) chunks_per_cloud = 1;
  chunk = (len + chunks_per_cloud - 1) / chunks_per_cloud;
  chunk = (chunk + kWgBK - 1) / kWgBK * kWgBK;       // whole slabs
  if (chunk < 4 * kWgBK) chunk = 4 * kWgBK;
  chunks_per_cloud = (len + chunk - 1) / chunk;
}

}  // namespace mvp

using namespace mvp;

extern "C" long long mvp_pointwise_wgrad_mfma_scratch_bytes(int b, int cin, int cout, int len, int with_bias) {
  if (b <= 0 || cin <= 0 || cout <= 0 || len <= 0 || (len & 3) != 0) return 0;
  int chunk, cpc, tiles;
  wgrad_plan(b, cin, cout, len, with_bias, chunk, cpc, tiles);
  const long long splits = (long long)b * cpc;
  if (splits > 65535 || tiles > 2147483647) return 0;
  return splits * cout * (cin + (with_bias ? 1 : 0)) * 4;
}

extern "C" int mvp_pointwise_wgrad_mfma(int b, int cin, int cout, int len, const float *x, const float *gy,
                                        const float *gymask, float *gw, float *gb, void *scratch,
                                        long long scratch_bytes, void *stream) {
  const int with_bias = gb != nullptr;
  const long long need = mvp_pointwise_wgrad_mfma_scratch_bytes(b, cin, cout, len, with_bias);
  if (need == 0) return MVP_EBADSHAPE;
  if (!x || !gy || !gw || !scratch || scratch_bytes < need) return MVP_EBADARG;
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(gymask)) & 15) != 0)
    return MVP_EBADARG;
  int chunk, cpc, tiles;
  wgrad_plan(b, cin, cout, len, with_bias, chunk, cpc, tiles);
  const int splits = b * cpc;
  hipStream_t st = as_stream(stream);
  float *partial = static_cast<float *>(scratch);
  if (cout > 64)
    hipLaunchKernelGGL(pointwise_wgrad_mfma_kernel<128>, dim3(tiles, splits), dim3(kMmThreads), 0, st, b, cin, cout, len,
                       chunk, cpc, x, gy, gymask, with_bias, partial);
  else
    hipLaunchKernelGGL(pointwise_wgrad_mfma_kernel<64>, dim3(tiles, splits), dim3(kMmThreads), 0, st, b, cin, cout, len,
                       chunk, cpc, x, gy, gymask, with_bias, partial);
  const long long total = (long long)cout * (cin + with_bias);
  hipLaunchKernelGGL(pointwise_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, cin, cout,
                     with_bias, splits, partial, gw, gb);
  return check_launch("mvp_pointwise_wgrad_mfma");
}

extern "C" int mvp_pointwise_mfma(int b, int cin, int cout, int len, const float *x, const float *xmask,
                                  const float *w, int w_kmajor, const float *bias, const float *residual, int relu,
                                  int group, float *y, void *stream) {
  if (b < 0 || cin <= 0 || cout <= 0 || len < 0) return MVP_EBADSHAPE;
  if (group < 1 || group > 32 || (group & (group - 1)) != 0) return MVP_EBADSHAPE;
  if ((len & 3) != 0 || len % group != 0 || b > 65535) return MVP_EBADSHAPE;
  if (b == 0 || len == 0) return MVP_OK;
  if (!x || !w || !y) return MVP_EBADARG;
  if ((reinterpret_cast<uintptr_t>(x) & 15) != 0 || (reinterpret_cast<uintptr_t>(xmask) & 15) != 0) return MVP_EBADARG;
  if (w_kmajor && ((cout & 3) != 0 || (reinterpret_cast<uintptr_t>(w) & 15) != 0)) return MVP_EBADARG;
  if (!w_kmajor && (cin & 3) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) != 0) return MVP_EBADARG;
  hipStream_t st = as_stream(stream);
  const bool big = cout > 64;
  dim3 grid((len + kMmBN - 1) / kMmBN, (cout + (big ? 128 : 64) - 1) / (big ? 128 : 64), b);
  if (grid.y > 65535) return MVP_EBADSHAPE;
#define MVP_MM(BM, KM)                                                                                       \
  hipLaunchKernelGGL((pointwise_mfma_kernel<BM, KM>), grid, dim3(kMmThreads), 0, st, cin, cout, len, x, xmask, w, \
                     bias, residual, relu, group, y)
  if (big) {
    if (w_kmajor) MVP_MM(128, true);
    else MVP_MM(128, false);
  } else {
    if (w_kmajor) MVP_MM(64, true);
    else MVP_MM(64, false);
  }
#undef MVP_MM
  return check_launch("mvp_pointwise_mfma");
}
