// Furthest point sampling for gfx950.
//
// Replaces furthest_point_sampling_kernel<BS> and
// furthest_point_sampling_with_dist_kernel<BS>
// (utils/mm3d_pn2/ops/furthest_point_sample/src/
//  furthest_point_sample_cuda.cu:26-141, 214-330).
//
// FPS is a latency chain: m-1 strictly sequential rounds, each = {update the
// running min-distance of every point to the sampled set, arg-max}.  The
// reference re-reads xyz (12 B) and temp (4 B r/w) of all N points from
// global memory every round and runs an 11-barrier shared-memory tree.
// MI355X-first design:
//   * one workgroup per cloud with the reference's own block size BS
//     (opt_n_threads, :11-15), so lane t owns exactly the points
//     k = t, t+BS, ... the reference's thread t scans;
//   * those points AND their running min-distance live in REGISTERS for the
//     whole kernel (P = ceil(N/BS) <= 16 points per lane => N <= 16384):
//     per round there is no global/LDS traffic for point data at all, HBM is
//     touched once (12 B/point in, 4 B/sample out);
//   * the arg-max is ONE packed 64-bit max reduction
//       key = float_bits(best) << 32 | (BS-1 - bitrev(t)) << 20 | k
//     done with wave64 cross-lane ops and one LDS hop across the <=16 waves
//     (double-buffered by round parity => a single barrier per round instead
//     of ~11).  The key reproduces the reference's tie rule exactly: inside a
//     thread the first strict maximum in k order (:69-70); across threads the
//     tree `v2 > v1 ? i2 : i1` (:17-23) lets the slot with the smallest
//     bit-reversed thread id win among equal maxima.
//   * N > 16384 falls back to a streaming variant (temp in global memory).
#include <cmath>

#include "common.h"

namespace mvp {

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const unsigned long long o = __shfl_xor(v, off, kWave);
    v = o > v ? o : v;
  }
  return v;
}

// P > 0: register-resident (n <= P*BS).  P == 0: streaming fallback.
// WITH_DIST: dataset is a (N,N) distance matrix per cloud (F-FPS).
template <int P, bool WITH_DIST>
__global__ __launch_bounds__(1024) void fps_kernel(
    int n, int m, int log2bs, const float *__restrict__ dataset,
    float *__restrict__ temp, int *__restrict__ idxs) {
  if (m <= 0) return;
  const int bs = 1 << log2bs;  // reference block size; blockDim.x = max(bs, 64)
  const int t = threadIdx.x;
  const bool live = t < bs;  // lanes beyond bs only pad the wave
  const int lane = t & (kWave - 1);
  const int wave = t >> 6;
  const int nwaves = (bs + kWave - 1) / kWave;  // <= 16
  const int cloud = blockIdx.x;
  dataset += (size_t)cloud * n * (WITH_DIST ? (size_t)n : 3);
  temp += (size_t)cloud * n;
  idxs += (size_t)cloud * m;

  __shared__ unsigned long long wbest[2][16];

  // bit-reverse t within log2bs bits; smaller reversed id wins ties.
  const unsigned rev = log2bs ? (__brev((unsigned)t) >> (32 - log2bs)) : 0u;
  const unsigned long long tiekey =
      (unsigned long long)((unsigned)(bs - 1) - rev) << 20;

  float px[P > 0 ? P : 1], py[P > 0 ? P : 1], pz[P > 0 ? P : 1],
      pt[P > 0 ? P : 1];
  if (P > 0) {
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int k = t + i * bs;
      const bool valid = live && k < n;
      if (!WITH_DIST) {
        px[i] = valid ? dataset[(size_t)k * 3 + 0] : 0.f;
        py[i] = valid ? dataset[(size_t)k * 3 + 1] : 0.f;
        pz[i] = valid ? dataset[(size_t)k * 3 + 2] : 0.f;
      }
      // temp starts at 1e10 (furthest_point_sample.py:30); padding lanes get
      // -inf so they can never be selected.
      pt[i] = valid ? 1e10f : -__builtin_inff();
    }
  } else {
    if (live)
      for (int k = t; k < n; k += bs) temp[k] = 1e10f;
  }

  int old = 0;
  if (t == 0) idxs[0] = old;

  for (int j = 1; j < m; ++j) {
    float x1 = 0.f, y1 = 0.f, z1 = 0.f;
    if (!WITH_DIST) {
      x1 = dataset[(size_t)old * 3 + 0];
      y1 = dataset[(size_t)old * 3 + 1];
      z1 = dataset[(size_t)old * 3 + 2];
    }
    float best = -1.f;
    int besti = 0;
    if (P > 0) {
#pragma unroll
      for (int i = 0; i < P; ++i) {
        const int k = t + i * bs;
        float d;
        if (WITH_DIST) {
          d = k < n ? dataset[(size_t)old * n + k] : 0.f;
        } else {
          d = sqdist3(px[i] - x1, py[i] - y1, pz[i] - z1);
        }
        const float d2 = d < pt[i] ? d : pt[i];
        pt[i] = d2;
        const bool gt = d2 > best;
        besti = gt ? k : besti;
        best = gt ? d2 : best;
      }
    } else {
      for (int k = live ? t : n; k < n; k += bs) {
        float d;
        if (WITH_DIST) {
          d = dataset[(size_t)old * n + k];
        } else {
          d = sqdist3(dataset[(size_t)k * 3 + 0] - x1,
                      dataset[(size_t)k * 3 + 1] - y1,
                      dataset[(size_t)k * 3 + 2] - z1);
        }
        const float tk = temp[k];
        const float d2 = d < tk ? d : tk;
        temp[k] = d2;
        const bool gt = d2 > best;
        besti = gt ? k : besti;
        best = gt ? d2 : best;
      }
    }
    // best >= 0 here (every thread owns at least point t < n), so its bit
    // pattern orders like an unsigned integer.
    unsigned long long key =
        ((unsigned long long)__float_as_uint(best) << 32) | tiekey |
        (unsigned long long)(unsigned)besti;
    if (!live) key = 0;
    key = wave_max_u64(key);
    if (nwaves > 1) {
      if (lane == 0) wbest[j & 1][wave] = key;
      __syncthreads();
      unsigned long long v = wbest[j & 1][0];
      for (int w = 1; w < nwaves; ++w) {
        const unsigned long long o = wbest[j & 1][w];
        v = o > v ? o : v;
      }
      key = v;
    }
    old = __builtin_amdgcn_readfirstlane((int)(key & 0xFFFFFu));
    if (t == 0) idxs[j] = old;
  }

  if (P > 0) {
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int k = t + i * bs;
      if (live && k < n) temp[k] = pt[i];
    }
  }
}

template <bool WITH_DIST>
static int launch_fps(int b, int n, int m, const float *dataset, float *temp,
                      int *idx, hipStream_t stream) {
  // opt_n_threads, furthest_point_sample_cuda.cu:11-15 -- same expression.
  const int pow_2 = (int)(std::log(static_cast<double>(n)) / std::log(2.0));
  int log2bs = pow_2 < 0 ? 0 : pow_2;
  if (log2bs > 10) log2bs = 10;
  const int bs = 1 << log2bs;
  const int p = (n + bs - 1) / bs;
  dim3 grid(b), block(bs < kWave ? kWave : bs);
#define MVP_FPS_CASE(PP)                                                      \
  hipLaunchKernelGGL((fps_kernel<PP, WITH_DIST>), grid, block, 0, stream, n, m, \
                     log2bs, dataset, temp, idx)
  if (n >= (1 << 20)) {
    return MVP_EBADSHAPE;  // index field of the packed key is 20 bits
  } else if (p <= 1) {
    MVP_FPS_CASE(1);
  } else if (p <= 2) {
    MVP_FPS_CASE(2);
  } else if (p <= 3) {
    MVP_FPS_CASE(3);
  } else if (p <= 4) {
    MVP_FPS_CASE(4);
  } else if (p <= 6) {
    MVP_FPS_CASE(6);
  } else if (p <= 8) {
    MVP_FPS_CASE(8);
  } else if (p <= 12) {
    MVP_FPS_CASE(12);
  } else if (p <= 16) {
    MVP_FPS_CASE(16);
  } else {
    MVP_FPS_CASE(0);
  }
#undef MVP_FPS_CASE
  return MVP_OK;
}

}  // namespace mvp

using namespace mvp;

extern "C" int mvp_furthest_point_sampling(int b, int n, int m,
                                           const float *points, float *temp,
                                           int *idx, void *stream) {
  if (b < 0 || n <= 0 || m < 0) return MVP_EBADSHAPE;
  if (b == 0 || m == 0) return MVP_OK;
  if (!points || !temp || !idx) return MVP_EBADARG;
  int rc = launch_fps<false>(b, n, m, points, temp, idx, as_stream(stream));
  if (rc != MVP_OK) return rc;
  return check_launch("mvp_furthest_point_sampling");
}

extern "C" int mvp_furthest_point_sampling_with_dist(int b, int n, int m,
                                                     const float *points_dist,
                                                     float *temp, int *idx,
                                                     void *stream) {
  if (b < 0 || n <= 0 || m < 0) return MVP_EBADSHAPE;
  if (b == 0 || m == 0) return MVP_OK;
  if (!points_dist || !temp || !idx) return MVP_EBADARG;
  int rc = launch_fps<true>(b, n, m, points_dist, temp, idx, as_stream(stream));
  if (rc != MVP_OK) return rc;
  return check_launch("mvp_furthest_point_sampling_with_dist");
}
