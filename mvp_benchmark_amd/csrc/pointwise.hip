// Weight gradient of the models' per-point / per-edge linear maps (1x1
// convolutions) with few channels -- SURVEY 8f, the grouped-feature MLP rows.
//   gw[co][ci] = sum_{b,l} gy[b][co][l] * x[b][ci][l],   gb[co] = sum_{b,l} gy[b][co][l]
// with x (B, Cin, L), gy (B, Cout, L), L = N or k*N positions per cloud: a GEMM
// whose output is tiny (Cout x Cin <= 64 x 512) and whose reduction dimension is
// everything else.  MIOpen's weight-gradient kernels run these shapes at 0.4-1.0
// TB/s of their inputs (profiles/r1k); the forward and data-gradient passes stay
// MIOpen calls (1.3-2.7 TB/s).
//
// A workgroup owns a run of positions of one cloud and a block of CK = 4 CIG
// input channels.  Tiles of 64 positions of x and gy are staged in LDS with
// coalesced 16-byte loads (every input byte is read once per channel block); a
// thread owns a 4 x 4 block of (co, ci) pairs -- COG x CIG such threads cover
// the layer, and the 256 / (COG CIG) copies of that arrangement split the tile's
// positions between them -- and walks its share of the tile with 16-byte LDS
// reads (8 reads feed 64 multiply-adds).  The copies are added in order inside
// the workgroup, one partial per workgroup goes to scratch, and a second small
// kernel adds the partials in a fixed order (no float atomics: the result is
// deterministic).
#include "common.h"

namespace mvp {

constexpr int kPwThreads = 256;
constexpr int kPwTile = 64;              // positions per LDS tile
constexpr int kPwStride = kPwTile + 4;   // row stride in floats (16-byte aligned rows)
constexpr int kPwRun = 16;               // tiles per workgroup
constexpr int kPwMaxRows = 224;          // 4 (CIG + COG) LDS rows: 59.5 KiB

struct PwPlan {
  int cog, cig, pg, chunks;  // co-groups, ci-groups, position groups, blocks of input channels
};
static PwPlan pw_plan(int cin, int cout) {
  PwPlan p;
  p.cog = (cout + 3) / 4;  // <= 16
  p.cig = (cin + 3) / 4;
  const int cap = min(kPwThreads / p.cog, kPwMaxRows / 4 - p.cog);
  if (p.cig > cap) p.cig = cap;
  p.pg = kPwThreads / (p.cog * p.cig);
  if (p.pg > kPwTile / 4) p.pg = kPwTile / 4;  // at most one group per 4 positions
  p.chunks = (cin + 4 * p.cig - 1) / (4 * p.cig);
  return p;
}

__global__ __launch_bounds__(kPwThreads) void pointwise_wgrad_kernel(
    int cin, int cout, int len, int COG, int CIG, int PG, const float *__restrict__ x, const float *__restrict__ gy,
    float *__restrict__ partial_w, float *__restrict__ partial_b) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int CK = 4 * CIG, CO = 4 * COG;
  float *s_x = reinterpret_cast<float *>(smem);  // CK rows
  float *s_g = s_x + CK * kPwStride;             // CO rows
  const int t = threadIdx.x;
  const int pgid = t / (COG * CIG);              // >= PG: idle in the accumulation (still loads)
  const int cog = (t / CIG) % COG, cig = t % CIG;  // lanes of a wave: consecutive ci-groups
  const int ci0 = blockIdx.y * CK;
  const int cloud = blockIdx.z;
  const int nrun = (len + kPwTile * kPwRun - 1) / (kPwTile * kPwRun);
  const int p_begin = blockIdx.x * kPwTile * kPwRun;
  const int p_end = min(len, p_begin + kPwTile * kPwRun);
  const float *xb = x + (size_t)cloud * cin * len;
  const float *gb = gy + (size_t)cloud * cout * len;
  float acc[4][4];
  float bsum[4];  // threads with cig == 0: bias gradient of their 4 output channels (first channel block only)
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    bsum[a] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = 0.f;
  }
  for (int p0 = p_begin; p0 < p_end; p0 += kPwTile) {
    __syncthreads();  // the previous tile is no longer read
    // rows 0..CK-1: x, rows CK..CK+CO-1: gy; 16 float4 per row (len % 4 == 0)
    for (int e = t; e < (CK + CO) * (kPwTile / 4); e += kPwThreads) {
      const int row = e / (kPwTile / 4), q = e % (kPwTile / 4);
      const int p = p0 + q * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < CK) {
        if (ci0 + row < cin && p < len) v = *reinterpret_cast<const float4 *>(xb + (size_t)(ci0 + row) * len + p);
      } else {
        if (row - CK < cout && p < len) v = *reinterpret_cast<const float4 *>(gb + (size_t)(row - CK) * len + p);
      }
      *reinterpret_cast<float4 *>(s_x + row * kPwStride + q * 4) = v;  // s_g follows s_x row for row
    }
    __syncthreads();
    // thread (pgid, cog, cig): output channels cog + COG a, input channels cig + CIG c (interleaved rows:
    // neighbouring lanes read neighbouring LDS rows), positions 4 q .. 4 q + 3 for q = pgid, pgid + PG, ...
    if (pgid < PG) {
      for (int q = pgid; q < kPwTile / 4; q += PG) {
        float4 gv[4], xv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) gv[a] = *reinterpret_cast<const float4 *>(s_g + (cog + COG * a) * kPwStride + q * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) xv[c] = *reinterpret_cast<const float4 *>(s_x + (cig + CIG * c) * kPwStride + q * 4);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          bsum[a] += (gv[a].x + gv[a].y) + (gv[a].z + gv[a].w);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            acc[a][c] += gv[a].x * xv[c].x;
            acc[a][c] += gv[a].y * xv[c].y;
            acc[a][c] += gv[a].z * xv[c].z;
            acc[a][c] += gv[a].w * xv[c].w;
          }
        }
      }
    }
  }
  // The PG copies are added inside the workgroup, in copy order, through LDS:
  // one partial per workgroup leaves it.
  __syncthreads();
  float *s_red = reinterpret_cast<float *>(smem);  // 16 sums per thread, then 4 bias sums per thread
  float *s_rb = s_red + kPwThreads * 16;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int c = 0; c < 4; ++c) s_red[t * 16 + a * 4 + c] = acc[a][c];
    s_rb[t * 4 + a] = bsum[a];
  }
  __syncthreads();
  const size_t blk = (size_t)cloud * nrun + blockIdx.x;
  float *pw = partial_w + blk * cout * cin;
  const int group = COG * CIG;
  for (int i = t; i < CO * CK; i += kPwThreads) {
    const int co = i / CK, r = i % CK;
    const int ci = ci0 + r;
    if (co < cout && ci < cin) {
      const int src = ((co % COG) * CIG + (r % CIG)) * 16 + (co / COG) * 4 + (r / CIG);
      float v = 0.f;
      for (int g = 0; g < PG; ++g) v += s_red[g * group * 16 + src];
      pw[(size_t)co * cin + ci] = v;
    }
  }
  if (blockIdx.y == 0 && t < cout) {
    const int src = ((t % COG) * CIG) * 4 + t / COG;
    float v = 0.f;
    for (int g = 0; g < PG; ++g) v += s_rb[g * group * 4 + src];
    partial_b[blk * cout + t] = v;
  }
}

// out[e] = sum over the nblk partials: 64 threads per element take every 64th
// partial in order and are combined by a fixed-shape butterfly (deterministic).
__global__ __launch_bounds__(256) void pointwise_wgrad_reduce_kernel(int nelem, int nblk, const float *__restrict__ partial,
                                                                     float *__restrict__ out) {
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  if (e < nelem)
    for (int k = lane; k < nblk; k += 64) s += partial[(size_t)k * nelem + e];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (e < nelem && lane == 0) out[e] = s;
}

// Data gradient of the same layers (cin, cout <= 64):  gx[b][ci][l] = sum_co w[co][ci] * gy[b][co][l].
// A thread owns two neighbouring positions and CI input channels (accumulators in registers), the
// weights are wave-uniform operands; gy is read once (8 B per lane and output channel, coalesced),
// gx written once: the op is bound by those 4 (cin + cout) bytes per position.  The library's
// implicit-GEMM kernel for these shapes wraps the same arithmetic in NCHW <-> NHWC transposes -- and
// one of its variants reads out of bounds (a cold-cache MIOpen picked
// igemm_bwd_gtcx35_nhwc_fp32_bx0_ex1_bt256x64x4 for ECG's 24-channel edge convolutions and faulted,
// depending on what the allocator had mapped behind the tensor; rocgdb trace in profiles/NOTES_r1-r3_design_notebook.md section 10).
template <int CI>
__global__ __launch_bounds__(256) void pointwise_dgrad_kernel(int cin, int cout, int len, const float *__restrict__ w,
                                                               const float *__restrict__ gy, float *__restrict__ gx) {
  // the weights, zero-padded to rows of 64: every lane reads the same words (LDS broadcast, 16 bytes per read)
  __shared__ float4 sw[64 * 16];
  for (int i = threadIdx.x; i < cout * 64; i += 256) {
    const int co = i >> 6, ci = i & 63;
    reinterpret_cast<float *>(sw)[i] = ci < cin ? w[(size_t)co * cin + ci] : 0.f;
  }
  __syncthreads();
  const int b = blockIdx.y;
  const int l = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (l >= len) return;
  gy += (size_t)b * cout * len + l;
  gx += (size_t)b * cin * len + l;
  for (int c0 = 0; c0 < cin; c0 += CI) {
    float2 acc[CI];
#pragma unroll
    for (int i = 0; i < CI; ++i) acc[i] = make_float2(0.f, 0.f);
    float2 g = *reinterpret_cast<const float2 *>(gy);
    for (int co = 0; co < cout; ++co) {
      const float2 gn = co + 1 < cout ? *reinterpret_cast<const float2 *>(gy + (size_t)(co + 1) * len) : g;   // next row in flight
#pragma unroll
      for (int i = 0; i < CI; i += 4) {
        const float4 w4 = sw[co * 16 + ((c0 + i) >> 2)];
        acc[i + 0].x = __builtin_fmaf(w4.x, g.x, acc[i + 0].x); acc[i + 0].y = __builtin_fmaf(w4.x, g.y, acc[i + 0].y);
        acc[i + 1].x = __builtin_fmaf(w4.y, g.x, acc[i + 1].x); acc[i + 1].y = __builtin_fmaf(w4.y, g.y, acc[i + 1].y);
        acc[i + 2].x = __builtin_fmaf(w4.z, g.x, acc[i + 2].x); acc[i + 2].y = __builtin_fmaf(w4.z, g.y, acc[i + 2].y);
        acc[i + 3].x = __builtin_fmaf(w4.w, g.x, acc[i + 3].x); acc[i + 3].y = __builtin_fmaf(w4.w, g.y, acc[i + 3].y);
      }
      g = gn;
    }
#pragma unroll
    for (int i = 0; i < CI; ++i)
      if (c0 + i < cin) *reinterpret_cast<float2 *>(gx + (size_t)(c0 + i) * len) = acc[i];
  }
}

static long long pw_blocks(int b, int len) { return (long long)b * ((len + kPwTile * kPwRun - 1) / (kPwTile * kPwRun)); }

}  // namespace mvp

using namespace mvp;

extern "C" long long mvp_pointwise_wgrad_scratch_bytes(int b, int cin, int cout, int len) {
  if (b <= 0 || cin <= 0 || cout <= 0 || cout > 64 || len <= 0 || len % 4 != 0 || b > 65535) return 0;
  const PwPlan pl = pw_plan(cin, cout);
  if (pl.chunks > 65535) return 0;
  return pw_blocks(b, len) * ((long long)cout * cin + cout) * 4;
}

extern "C" int mvp_pointwise_dgrad(int b, int cin, int cout, int len, const float *weight, const float *gy, float *gx,
                                   void *stream) {
  if (b <= 0 || cin <= 0 || cout <= 0 || cin > 64 || cout > 64 || len <= 0 || len % 4 != 0 || b > 65535) return MVP_EBADSHAPE;
  if (!weight || !gy || !gx) return MVP_EBADARG;
  if (((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(gx)) & 15) != 0) return MVP_EBADARG;
  hipStream_t st = as_stream(stream);
  const dim3 grid((len / 2 + 255) / 256, b), block(256);
  // (one pass over gy when all input channels fit the accumulators; 64 channels take two)
  if (cin <= 8) hipLaunchKernelGGL(pointwise_dgrad_kernel<8>, grid, block, 0, st, cin, cout, len, weight, gy, gx);
  else if (cin <= 16) hipLaunchKernelGGL(pointwise_dgrad_kernel<16>, grid, block, 0, st, cin, cout, len, weight, gy, gx);
  else if (cin <= 24) hipLaunchKernelGGL(pointwise_dgrad_kernel<24>, grid, block, 0, st, cin, cout, len, weight, gy, gx);
  else if (cin <= 48) hipLaunchKernelGGL(pointwise_dgrad_kernel<48>, grid, block, 0, st, cin, cout, len, weight, gy, gx);
  else hipLaunchKernelGGL(pointwise_dgrad_kernel<32>, grid, block, 0, st, cin, cout, len, weight, gy, gx);
  return check_launch("mvp_pointwise_dgrad");
}

extern "C" int mvp_pointwise_wgrad(int b, int cin, int cout, int len, const float *x, const float *gy, float *gw,
                                   float *gb, void *scratch, long long scratch_bytes, void *stream) {
  const long long need = mvp_pointwise_wgrad_scratch_bytes(b, cin, cout, len);
  if (need == 0) return MVP_EBADSHAPE;
  if (!x || !gy || !gw || !scratch || scratch_bytes < need) return MVP_EBADARG;
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy)) & 15) != 0) return MVP_EBADARG;
  const PwPlan pl = pw_plan(cin, cout);
  const int nrun = (len + kPwTile * kPwRun - 1) / (kPwTile * kPwRun);
  const long long npart = pw_blocks(b, len);
  if (npart > 2147483647LL) return MVP_EBADSHAPE;
  float *pw = static_cast<float *>(scratch);
  float *pb = pw + npart * cout * cin;
  hipStream_t st = as_stream(stream);
  size_t lds = (size_t)4 * (pl.cig + pl.cog) * kPwStride * sizeof(float);
  if (lds < (size_t)kPwThreads * 20 * sizeof(float)) lds = (size_t)kPwThreads * 20 * sizeof(float);  // the closing reduction
  hipLaunchKernelGGL(pointwise_wgrad_kernel, dim3(nrun, pl.chunks, b), dim3(kPwThreads), lds, st, cin, cout, len, pl.cog,
                     pl.cig, pl.pg, x, gy, pw, pb);
  hipLaunchKernelGGL(pointwise_wgrad_reduce_kernel, dim3((cout * cin + 3) / 4), dim3(256), 0, st, cout * cin, (int)npart,
                     pw, gw);
  if (gb)
    hipLaunchKernelGGL(pointwise_wgrad_reduce_kernel, dim3((cout + 3) / 4), dim3(256), 0, st, cout, (int)npart, pb, gb);
  return check_launch("mvp_pointwise_wgrad");
}
