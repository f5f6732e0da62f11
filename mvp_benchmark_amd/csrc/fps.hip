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
//   * the hot loop only updates the running minima and their per-lane maximum
//     (no index tracking); the arg-max is ONE max reduction of the packed key
//       key = float_bits(best) << 32 | (BS-1 - bitrev(t)) << 20 | t
//     done with DPP row operations inside the wave -- as two 32-bit maxima
//     (the value, then the tie word among its holders; a single holder's tie
//     word is fetched with one v_readlane): one DPP-fused v_max_u32 per step
//     instead of a 64-bit compare-and-select chain -- and ONE ds_max_u64 per
//     wave on a double-buffered LDS word across the <=16 waves (a round is
//     bound by the ~80 instruction slots every wave spends on it, 4 waves to
//     a SIMD; the LDS atomic replaces a second reduction stage of ~25 slots).
//     Only the winning lane then resolves which of its points
//     holds the maximum and publishes index + coordinates from its registers
//     through LDS (2 barriers per round instead of ~11, no global read for the
//     selected point).  The key reproduces the reference's tie rule exactly:
//     inside a thread the first strict maximum in k order (:69-70); across
//     threads the tree `v2 > v1 ? i2 : i1` (:17-23) lets the slot with the
//     smallest bit-reversed thread id win among equal maxima.
//   * N > 16384 falls back to a streaming variant (temp in global memory).
#include <cmath>
#include <type_traits>

#include "common.h"

namespace mvp {

// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS
// operations, not for its outstanding global stores (__syncthreads() would also
// drain vmcnt, putting the L2 round trip of the per-round `idxs[j]` store on
// the critical path of every round).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// Unsigned 32-bit max with one DPP-modified operand: a single v_max_u32 per
// step (lanes a row mask leaves unwritten combine with 0, the identity).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_umax32(unsigned v) {
  const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
  return o > v ? o : v;
}
__device__ __forceinline__ unsigned wave_umax32(unsigned v) {
  v = dpp_umax32<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v = dpp_umax32<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v = dpp_umax32<0x141, 0xF>(v);  // row_half_mirror
  v = dpp_umax32<0x140, 0xF>(v);  // row_mirror
  v = dpp_umax32<0x142, 0xA>(v);  // row_bcast15 -> rows 1, 3
  v = dpp_umax32<0x143, 0xC>(v);  // row_bcast31 -> rows 2, 3
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned row16_umax32(unsigned v) {  // butterfly: every lane of the row gets the max
  v = dpp_umax32<0xB1, 0xF>(v);
  v = dpp_umax32<0x4E, 0xF>(v);
  v = dpp_umax32<0x141, 0xF>(v);
  v = dpp_umax32<0x140, 0xF>(v);
  return v;
}

// Wave64 max of a u64 key {value bits, tie word}, returned wave-uniform.  A
// 64-bit compare-and-select costs five dependent instructions per DPP step;
// the lexicographic max is instead taken as two 32-bit reductions -- the value,
// then the tie word among the lanes that hold that value (0 elsewhere).
// When a single lane holds the maximal value (every round on generic data) its
// tie word is fetched with one v_readlane instead of a second DPP reduction.
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
  const unsigned mh = wave_umax32(hi);
  const unsigned long long holders = __ballot(hi == mh);  // never empty
  unsigned ml;
  if ((holders & (holders - 1ull)) == 0ull)  // wave-uniform
    ml = (unsigned)__builtin_amdgcn_readlane((int)lo, __builtin_ctzll(holders));
  else
    ml = wave_umax32(hi == mh ? lo : 0u);
  return ((unsigned long long)mh << 32) | ml;
}
// The same over the 16 lanes of row 0 (valid in every lane of that row; taken
// from lane 0).
__device__ __forceinline__ unsigned long long row16_max_u64(unsigned long long v) {
  const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
  const unsigned mh = (unsigned)__builtin_amdgcn_readfirstlane((int)row16_umax32(hi));
  const unsigned long long holders = __ballot(hi == mh) & 0xFFFFull;  // row 0 only; never empty
  unsigned ml;
  if ((holders & (holders - 1ull)) == 0ull)
    ml = (unsigned)__builtin_amdgcn_readlane((int)lo, __builtin_ctzll(holders));
  else
    ml = (unsigned)__builtin_amdgcn_readfirstlane((int)row16_umax32(hi == mh ? lo : 0u));
  return ((unsigned long long)mh << 32) | ml;
}

// Squared distances of P points to (x1, y1, z1) -- sqdist3's chain fma(dz, dz, fma(dy, dy, dx * dx)) -- two points per
// instruction: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 work on a register pair with one result per component,
// rounded exactly as their scalar forms (three packed instructions per point instead of six; the CU's packed FP32
// rate is what the 157 TFLOP/s vector peak counts).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f sqdist3_pair(float xa, float xb, float ya, float yb, float za, float zb, float x1, float y1,
                                            float z1) {
  const v2f dx = v2f{xa, xb} - x1, dy = v2f{ya, yb} - y1, dz = v2f{za, zb} - z1;
  v2f m = dx * dx;
  m = __builtin_elementwise_fma(dy, dy, m);
  m = __builtin_elementwise_fma(dz, dz, m);
  return m;
}
// distance of point i of a lane's P: pairs (i, i + 1) share their instructions (the compiler merges the two calls of
// a pair); an odd last point takes the scalar chain
template <int P>
__device__ __forceinline__ float sqdist3_of(const float (&px)[P], const float (&py)[P], const float (&pz)[P], int i, float x1,
                                            float y1, float z1) {
  const int a = i & ~1;
  if (a + 1 < P) {
    const v2f m = sqdist3_pair(px[a], px[a + 1], py[a], py[a + 1], pz[a], pz[a + 1], x1, y1, z1);
    return (i & 1) ? m.y : m.x;
  }
  return sqdist3(px[i] - x1, py[i] - y1, pz[i] - z1);
}

// min(d, t) as ONE v_min_f32: no NaN ever enters here, so it equals the reference's `d < t ? d : t`
// (furthest_point_sample_cuda.cu:62-66 `min(d, temp[k])`); spelled in assembly because __builtin_fminf adds a
// canonicalising v_max_f32 in front (IEEE mode) and the select form costs v_cmp + wait state + v_cndmask.
__device__ __forceinline__ float fmin_raw(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// (x, y, z) of point `ci` of lane `hl` -- both wave-uniform -- out of the lanes' register arrays: a balanced tree of
// scalar branches down to three v_readlane (registers cannot be indexed: the alternative, carrying the arg-max
// point's coordinates through the update loop, costs three v_cndmask per point and three VGPRs)
template <int LO, int HI, int P>
__device__ __forceinline__ void pick_point(const float (&px)[P], const float (&py)[P], const float (&pz)[P], int ci, int hl,
                                           float &x, float &y, float &z) {
  if constexpr (HI - LO == 1) {
    x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px[LO]), hl));
    y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(py[LO]), hl));
    z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pz[LO]), hl));
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (ci < MID) pick_point<LO, MID, P>(px, py, pz, ci, hl, x, y, z);
    else pick_point<MID, HI, P>(px, py, pz, ci, hl, x, y, z);
  }
}

// wave_max_u64 that also reports whether more than one lane holds the maximal VALUE (high word)
__device__ __forceinline__ unsigned long long wave_max_u64_dup(unsigned long long v, bool &several) {
  const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
  const unsigned mh = wave_umax32(hi);
  const unsigned long long holders = __ballot(hi == mh);  // never empty
  several = (holders & (holders - 1ull)) != 0ull;
  unsigned ml;
  if (!several)  // wave-uniform
    ml = (unsigned)__builtin_amdgcn_readlane((int)lo, __builtin_ctzll(holders));
  else
    ml = wave_umax32(hi == mh ? lo : 0u);
  return ((unsigned long long)mh << 32) | ml;
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

  __shared__ unsigned long long s_max[2];  // block maximum of the packed key, by round parity
  __shared__ float s_sel[2][4];  // {old as int bits, x, y, z} of the selected point

  // bit-reverse t within log2bs bits; smaller reversed id wins ties.
  const unsigned rev = log2bs ? (__brev((unsigned)t) >> (32 - log2bs)) : 0u;
  // key = best bits << 32 | (bs-1-rev) << 20 | t  (t < 1024 < 2^20)
  const unsigned long long tiekey =
      ((unsigned long long)((unsigned)(bs - 1) - rev) << 20) | (unsigned)t;

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

  if (t < 2) s_max[t] = 0ull;
  __syncthreads();
  int old = 0;
  if (t == 0) idxs[0] = old;
  float x1 = 0.f, y1 = 0.f, z1 = 0.f;
  if (!WITH_DIST) {
    x1 = dataset[0];
    y1 = dataset[1];
    z1 = dataset[2];
  }

  // One round; PAR = j & 1 as a compile-time constant (the loop below alternates the two instances), so that the
  // double-buffer addressing folds into immediate offsets.
  auto round = [&](const int j, auto par_c) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value;
    float best = -1.f;
    int besti = 0;  // only tracked by the streaming variant
    if (P > 0) {
      // Update the running min-distances; the arg-max index is NOT tracked
      // here: only the winning lane resolves it afterwards.
#pragma unroll
      for (int i = 0; i < P; ++i) {
        const int k = t + i * bs;
        float d;
        if (WITH_DIST) {
          d = k < n ? dataset[(size_t)old * n + k] : 0.f;
        } else {
          d = sqdist3(px[i] - x1, py[i] - y1, pz[i] - z1);
        }
        pt[i] = d < pt[i] ? d : pt[i];
        best = __builtin_fmaxf(best, pt[i]);
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
    // Squared distances: best >= 0 here (every live thread owns at least point t < n), so its
    // bit pattern orders like an unsigned integer.  A distance MATRIX (F-FPS) may hold small
    // negative entries -- the reference builds it as |a|^2 + |b|^2 - 2ab in float32
    // (points_sampler.py:119-158, utils.py:4-31) -- so there the order-preserving map is used.
    unsigned bbits = __float_as_uint(best);
    if (WITH_DIST) bbits = (bbits & 0x80000000u) ? ~bbits : (bbits | 0x80000000u);
    unsigned long long key = ((unsigned long long)bbits << 32) | tiekey;
    if (!live) key = 0;
    key = wave_max_u64(key);
    if (nwaves > 1) {
      // <= 16 wave maxima -> block maximum: one ds_max_u64 per wave on a
      // double-buffered LDS word, one broadcast read after the barrier (no
      // second reduction stage); the other word is cleared for the next round.
      if (lane == 0) atomicMax(&s_max[PAR], key);
      lds_barrier();
      key = s_max[PAR];
      if (t == 0) s_max[1 - PAR] = 0ull;
    }
    const int tstar = (int)(key & 0xFFFFFu);  // winning thread
    if (wave == (tstar >> 6)) {               // wave-uniform
      // The winning lane resolves its first (lowest k) maximum -- the
      // reference's per-thread strict `>` scan (:69-70) -- and publishes the
      // selected point from its registers.
      unsigned vbits = (unsigned)(key >> 32);
      if (WITH_DIST) vbits = (vbits & 0x80000000u) ? (vbits ^ 0x80000000u) : ~vbits;
      const float vbest = __uint_as_float(vbits);
      int sel = P > 0 ? t : besti;
      float sx = 0.f, sy = 0.f, sz = 0.f;
      if (P > 0) {
#pragma unroll
        for (int i = P - 1; i >= 0; --i) {
          const bool hit = pt[i] == vbest;
          sel = hit ? t + i * bs : sel;
          if (!WITH_DIST) {
            sx = hit ? px[i] : sx;
            sy = hit ? py[i] : sy;
            sz = hit ? pz[i] : sz;
          }
        }
      }
      if (t == tstar) {
        s_sel[PAR][0] = __int_as_float(sel);
        s_sel[PAR][1] = sx;
        s_sel[PAR][2] = sy;
        s_sel[PAR][3] = sz;
      }
    }
    lds_barrier();
    old = __builtin_amdgcn_readfirstlane(__float_as_int(s_sel[PAR][0]));
    if (!WITH_DIST) {
      if (P > 0) {
        x1 = s_sel[PAR][1];
        y1 = s_sel[PAR][2];
        z1 = s_sel[PAR][3];
      } else {
        x1 = dataset[(size_t)old * 3 + 0];
        y1 = dataset[(size_t)old * 3 + 1];
        z1 = dataset[(size_t)old * 3 + 2];
      }
    }
    if (t == 0) idxs[j] = old;
  };
  {
    int j = 1;
    for (; j + 1 < m; j += 2) {
      round(j, std::integral_constant<int, 1>{});
      round(j + 1, std::integral_constant<int, 0>{});
    }
    if (j < m) round(j, std::integral_constant<int, 1>{});
  }

  if (P > 0) {
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int k = t + i * bs;
      if (live && k < n) temp[k] = pt[i];
    }
  }
}

// ---------------------------------------------------------------------------
// Spatially sorted variant for the large, VALU-bound shapes (4096 < N <= 16384,
// where every lane owns up to 16 points and a round costs 8 instructions per
// point for the distance update alone).
//
// fps_sort_kernel buckets the cloud into 16^3 Morton-ordered cells (counting
// sort in LDS) and writes {x, y, z, bits(original index)} in that order into
// caller scratch.  fps_sorted_kernel gives every lane P CONSECUTIVE sorted
// points -- a small box -- and skips the update of a whole wave when the new
// sample is, for each of its lanes, farther from the lane's box than the
// lane's largest running minimum (no point can change: min(temp, d) = temp;
// the box distance uses the monotone subtract/fma chain of sqdist3 on per-axis
// gaps, so it never exceeds a member's computed distance).  After ~100 samples
// 15-35 % of the waves still update.
// Lanes are no longer the reference's threads, so the arg-max key cannot carry
// the reference's tie order.  The reduction finds the maximum and one holder;
// a round in which the maximum is attained more than once (inside the holder's
// lane or by another lane; never on random data, every round on a lattice) is
// detected with one compare + ballot and resolved exactly from the ORIGINAL
// indices -- smallest bit-reversed slot k % BS, then smallest k
// (furthest_point_sample_cuda.cu:17-23,69-70) -- out of line, through LDS.
constexpr int kFsCells = 4096;

__global__ __launch_bounds__(1024) void fps_sort_kernel(int n, int npad, const float *__restrict__ xyz,
                                                        float4 *__restrict__ sorted) {
  const int cloud = blockIdx.x;
  const float *__restrict__ in = xyz + (size_t)cloud * n * 3;
  float4 *__restrict__ out = sorted + (size_t)cloud * npad;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  __shared__ int s_cnt[kFsCells];
  __shared__ int s_start[kFsCells];
  __shared__ float s_red[6][16];
  __shared__ int s_wsum[16];
  float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
  float mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  for (int k = t; k < n; k += 1024) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = in[k * 3 + a];
      mn[a] = __builtin_fminf(mn[a], v);
      mx[a] = __builtin_fmaxf(mx[a], v);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      mn[a] = __builtin_fminf(mn[a], __shfl_xor(mn[a], off, 64));
      mx[a] = __builtin_fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
    }
    if (lane == 0) {
      s_red[a][wave] = mn[a];
      s_red[3 + a][wave] = mx[a];
    }
  }
  for (int c = t; c < kFsCells; c += 1024) s_cnt[c] = 0;
  __syncthreads();
  float lo[3], ext = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float l = s_red[a][0], h = s_red[3 + a][0];
    for (int w = 1; w < 16; ++w) {
      l = __builtin_fminf(l, s_red[a][w]);
      h = __builtin_fmaxf(h, s_red[3 + a][w]);
    }
    lo[a] = l;
    ext = __builtin_fmaxf(ext, h - l);
  }
  if (!(ext > 0.f) || !(ext < 3.0e38f)) ext = 1.f;
  const float invh = 16.f / ext;
  auto spread4 = [](int v) {  // bit i -> bit 3i
    v &= 0xF;
    v = (v | (v << 4)) & 0xC3;
    v = (v | (v << 2)) & 0x249;
    return v;
  };
  auto cell_of = [&](float x, float y, float z) {
    const int ix = min(15, max(0, (int)((x - lo[0]) * invh)));
    const int iy = min(15, max(0, (int)((y - lo[1]) * invh)));
    const int iz = min(15, max(0, (int)((z - lo[2]) * invh)));
    return spread4(ix) | (spread4(iy) << 1) | (spread4(iz) << 2);
  };
  for (int k = t; k < n; k += 1024) atomicAdd(&s_cnt[cell_of(in[k * 3 + 0], in[k * 3 + 1], in[k * 3 + 2])], 1);
  __syncthreads();
  {
    int v[4], sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i] = s_cnt[4 * t + i];
      sum += v[i];
    }
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int w = 0; w < wave; ++w) base += s_wsum[w];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s_start[4 * t + i] = base;
      base += v[i];
    }
    __syncthreads();
    for (int c = t; c < kFsCells; c += 1024) s_cnt[c] = 0;
    __syncthreads();
  }
  for (int k = t; k < n; k += 1024) {
    const float x = in[k * 3 + 0], y = in[k * 3 + 1], z = in[k * 3 + 2];
    const int c = cell_of(x, y, z);
    out[s_start[c] + atomicAdd(&s_cnt[c], 1)] = make_float4(x, y, z, __int_as_float(k));
  }
  for (int k = n + t; k < npad; k += 1024) out[k] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
}

// Tie round, out of line: every lane has copied its running minima to s_pt; the
// lanes that hold the maximum rank their points by the reference's order on the
// original indices; returns {thread << 4 | point} of the winner to every lane.
template <int P, int NT>
__device__ __noinline__ unsigned fps_resolve_tie(const float (*s_pt)[NT], const int (*s_pk)[NT],
                                                 unsigned long long (*wbest)[16], int slot, float vbest) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  unsigned tk = 0u;  // {1023 - rev(slot), 1023 - k / 1024, i}: 10 + 10 + 4 bits
  for (int i = 0; i < P; ++i) {
    const int pki = s_pk[i][t];
    if (s_pt[i][t] == vbest && pki >= 0) {
      const unsigned k = (unsigned)pki;
      const unsigned rev = __brev(k & 1023u) >> 22;
      const unsigned cand = ((1023u - rev) << 14) | ((1023u - (k >> 10)) << 4) | (unsigned)i;
      tk = cand > tk ? cand : tk;
    }
  }
  unsigned long long k2 = wave_max_u64(((unsigned long long)tk << 10) | (unsigned)t);
  if (lane == 0) wbest[slot][wave] = k2;
  lds_barrier();
  const unsigned r = (unsigned)row16_max_u64(wbest[slot][lane & 15]);
  return ((r & 0x3FFu) << 4) | ((r >> 10) & 15u);
}

template <int P>
__global__ __launch_bounds__(1024) void fps_sorted_kernel(int n, int m, const float *__restrict__ dataset,
                                                          const float4 *__restrict__ sorted,
                                                          float *__restrict__ temp, int *__restrict__ idxs) {
  // (16 waves per cloud.  Round 5 also ran this kernel on 2 / 4 / 8 waves with P up to 32 points per lane -- lanes here
  // are not the reference's threads, so the block size is free -- to pay the per-wave cost of a round fewer times: slower
  // everywhere, e.g. (64, 2048 -> 2048) 1.41-1.80 ms against 1.12 for fps_kernel<2>: a round is bound by the dependent
  // chain of ONE wave, which grows with the points a lane owns; profiles/r5_fps_experiments.txt)
  constexpr int NW = 16;
  constexpr int NT = 64 * NW;
  if (m <= 0) return;
  const int t = threadIdx.x, lane = t & 63;
  const int cloud = blockIdx.x;
  dataset += (size_t)cloud * n * 3;
  sorted += (size_t)cloud * (NT * P);
  temp += (size_t)cloud * n;
  idxs += (size_t)cloud * m;
  __shared__ unsigned long long wbest[2][16];
  __shared__ float s_sel[2][4];   // tie rounds: the resolved point
  // every wave's candidate of the round, by round parity: {value bits << 32 | several holders << 10 | a holder thread}
  // and that holder's point {original index bits, x, y, z} -- ONE barrier per round (see fps_kernel's round_reg)
  __shared__ unsigned long long s_wkey[2][16];
  __shared__ float4 s_wsel[2][16];
  // original indices ([i][t]: conflict-free): read only by the publishing lane
  // and in tie rounds; s_pt receives the running minima in tie rounds
  __shared__ int s_pk[P][NT];
  __shared__ float s_pt[P][NT];

  float px[P], py[P], pz[P], pt[P];
  float blo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
  float bhi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const float4 p = sorted[t * P + i];
    px[i] = p.x; py[i] = p.y; pz[i] = p.z;
    const int pki = __float_as_int(p.w);
    s_pk[i][t] = pki;
    const bool valid = pki >= 0;
    pt[i] = valid ? 1e10f : -__builtin_inff();  // furthest_point_sample.py:30; padding can never win
    blo[0] = __builtin_fminf(blo[0], valid ? p.x : blo[0]); bhi[0] = __builtin_fmaxf(bhi[0], valid ? p.x : bhi[0]);
    blo[1] = __builtin_fminf(blo[1], valid ? p.y : blo[1]); bhi[1] = __builtin_fmaxf(bhi[1], valid ? p.y : bhi[1]);
    blo[2] = __builtin_fminf(blo[2], valid ? p.z : blo[2]); bhi[2] = __builtin_fmaxf(bhi[2], valid ? p.z : bhi[2]);
  }
  float lane_max = pt[0];
#pragma unroll
  for (int i = 1; i < P; ++i) lane_max = __builtin_fmaxf(lane_max, pt[i]);
  bool lane_dup = true;  // all valid points start equal
  // the lane's arg-max point, carried from the last update of its wave (every wave with a valid point updates in
  // round 1: every running minimum starts at 1e10)
  int ci = 0;
  int cpk = __float_as_int(sorted[t * P].w);

  if (t == 0) idxs[0] = 0;
  if (t < 32) wbest[t >> 4][t & 15] = 0ull;   // (tie rounds reduce all 16 slots; only NW are ever written)
  float x1 = dataset[0], y1 = dataset[1], z1 = dataset[2];
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // (uniform: the LDS addresses below are scalar work)
  __syncthreads();

  for (int j = 1; j < m; ++j) {
    const int par = j & 1;
    const float gx = __builtin_fmaxf(__builtin_fmaxf(blo[0] - x1, x1 - bhi[0]), 0.f);
    const float gy = __builtin_fmaxf(__builtin_fmaxf(blo[1] - y1, y1 - bhi[1]), 0.f);
    const float gz = __builtin_fmaxf(__builtin_fmaxf(blo[2] - z1, z1 - bhi[2]), 0.f);
    const bool need = sqdist3(gx, gy, gz) < lane_max;  // a lane of padding only: -inf, never
    if (__any(need)) {
      float best = -__builtin_inff(), second = -__builtin_inff();
      ci = 0;
#pragma unroll
      for (int i = 0; i < P; ++i) {
        const float d = sqdist3_of<P>(px, py, pz, i, x1, y1, z1);
        pt[i] = fmin_raw(d, pt[i]);
        const bool gt = pt[i] > best;    // strict: the first of equal maxima (a tie round re-ranks them anyway)
        second = __builtin_amdgcn_fmed3f(best, second, pt[i]);   // = max(second, min(best, pt[i])) as best >= second
        best = __builtin_fmaxf(best, pt[i]);
        ci = gt ? i : ci;
      }
      lane_max = best;
      lane_dup = second == best;  // the two largest coincide
      cpk = s_pk[ci][t];
    }
    // lane_max >= 0 for a lane with a valid point (bits order like unsigned);
    // lanes of padding contribute the smallest key
    unsigned long long key = lane_max >= 0.f ? ((unsigned long long)__float_as_uint(lane_max) << 32) | (unsigned)t : 0ull;
    bool several;
    key = wave_max_u64_dup(key, several);
    const int hl = (int)(key & 63ull);   // a lane that holds the wave's maximum
    several |= __builtin_amdgcn_readlane((int)lane_dup, hl) != 0;
    {
      float hx, hy, hz;
      pick_point<0, P, P>(px, py, pz, __builtin_amdgcn_readlane(ci, hl), hl, hx, hy, hz);
      const int hpk = __builtin_amdgcn_readlane(cpk, hl);
      if (lane == 0) {
        s_wkey[par][wave] = key | (several ? 1024ull : 0ull);
        s_wsel[par][wave] = make_float4(__int_as_float(hpk), hx, hy, hz);
      }
    }
    lds_barrier();
    // lanes 0..15 of every wave take one wave's candidate each; the maximal value names the winner; the maximum is
    // attained once only iff one wave holds it and that wave saw one holder with one maximal point
    const unsigned long long kw = (lane & 15) < NW ? s_wkey[par][lane & 15] : 0ull;
    const float4 cw = s_wsel[par][(lane & 15) < NW ? (lane & 15) : 0];
    const unsigned hi = (unsigned)(kw >> 32);
    const unsigned mh = (unsigned)__builtin_amdgcn_readfirstlane((int)row16_umax32(hi));
    const unsigned long long hold = __ballot(hi == mh) & 0xFFFFull;   // row 0: one lane per wave of the block
    const int ww = (int)__builtin_ctzll(hold);
    const unsigned lo_w = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)kw, ww);
    const float vbest = __uint_as_float(mh);
    int old;
    if (__builtin_expect((hold & (hold - 1ull)) != 0ull || (lo_w & 1024u) != 0u, 0)) {   // block-uniform: a tie round
#pragma unroll
      for (int i = 0; i < P; ++i) s_pt[i][t] = pt[i];
      const unsigned r = fps_resolve_tie<P, NT>(s_pt, s_pk, wbest, par, vbest);
      const int tstar = (int)(r >> 4);
      const int istar = (int)(r & 15u);
      if (t == tstar) {
        float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
        for (int i = 0; i < P; ++i) {
          const bool hit = i == istar;
          sx = hit ? px[i] : sx;
          sy = hit ? py[i] : sy;
          sz = hit ? pz[i] : sz;
        }
        s_sel[par][0] = __int_as_float(s_pk[istar][t]);
        s_sel[par][1] = sx;
        s_sel[par][2] = sy;
        s_sel[par][3] = sz;
      }
      lds_barrier();
      old = __builtin_amdgcn_readfirstlane(__float_as_int(s_sel[par][0]));
      x1 = s_sel[par][1];
      y1 = s_sel[par][2];
      z1 = s_sel[par][3];
      lds_barrier();   // (s_sel[par] may be rewritten by a tie two rounds on; wbest likewise)
    } else {
      old = __builtin_amdgcn_readlane(__float_as_int(cw.x), ww);
      x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cw.y), ww));
      y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cw.z), ww));
      z1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cw.w), ww));
    }
    if (t == 0) idxs[j] = old;
  }
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int pki = s_pk[i][t];
    if (pki >= 0) temp[pki] = pt[i];
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

// ---------------------------------------------------------------------------
// W workgroups per cloud (build it and measure it).
//
// The register-resident kernel gives a cloud ONE workgroup: 64 clouds occupy 64 of 256 CUs and a
// lane of a 16384-point cloud owns 16 points.  Here W = 2 or 4 workgroups share a cloud; lane t of
// member w owns the points k = t + (w + W i) * 1024, i.e. a subset of what the reference's thread
// t scans, so the per-lane work shrinks by W.  Per round every member finds its local maximum as
// fps_kernel does, its winning lane resolves the point and publishes {key, x, y, z} as five tagged
// 8-byte granules (one store each, write-through; the data is the flag, slots double-buffered by
// round parity -- the exchange of emd.hip's cluster barrier), every member polls all W x 5 words
// and takes the maximum of
//     key = value bits << 32 | (1023 - bitrev(t)) << 14 | (16383 - k)
// = the reference's winner: largest value, then smallest bit-reversed thread slot, then (inside a
// thread) the lowest k.  Cooperative launch (the members wait for each other).
// Measured (profiles/r3_fps_cluster.txt): see profiles/NOTES_r1-r3_design_notebook.md section 10.
constexpr unsigned kFpsSpinLimit = 1u << 24;

template <int PW, int W>
__global__ __launch_bounds__(1024) void fps_cluster_kernel(
    int b, int bpad, int n, int m, const float *__restrict__ dataset, float *__restrict__ temp,
    int *__restrict__ idxs, unsigned long long *__restrict__ granules) {
  const int cloud = (int)blockIdx.x % bpad;
  const int w = (int)blockIdx.x / bpad;
  if (cloud >= b || m <= 0) return;
  constexpr int bs = 1024;
  const int t = threadIdx.x;
  const int lane = t & (kWave - 1);
  const int wave = t >> 6;
  dataset += (size_t)cloud * n * 3;
  temp += (size_t)cloud * n;
  idxs += (size_t)cloud * m;
  unsigned long long *slots = granules + (size_t)cloud * (2 * W * 8);   // [parity][member][8 words, 5 used]

  __shared__ unsigned long long s_max[2];
  __shared__ float s_sel[2][4];
  __shared__ int s_abort;

  const unsigned rev = __brev((unsigned)t) >> 22;   // 10 bits
  const unsigned long long tiekey = ((unsigned long long)(1023u - rev) << 20) | (unsigned)t;

  float px[PW], py[PW], pz[PW], pt[PW];
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int k = t + (w + W * i) * bs;
    const bool valid = k < n;
    px[i] = valid ? dataset[(size_t)k * 3 + 0] : 0.f;
    py[i] = valid ? dataset[(size_t)k * 3 + 1] : 0.f;
    pz[i] = valid ? dataset[(size_t)k * 3 + 2] : 0.f;
    pt[i] = valid ? 1e10f : -__builtin_inff();
  }
  if (t < 2) s_max[t] = 0ull;
  if (t == 0) s_abort = 0;
  __syncthreads();
  int old = 0;
  if (w == 0 && t == 0) idxs[0] = 0;
  float x1 = dataset[0], y1 = dataset[1], z1 = dataset[2];

  for (int j = 1; j < m; ++j) {
    const int par = j & 1;
    float best = -__builtin_inff();
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const float d = sqdist3(px[i] - x1, py[i] - y1, pz[i] - z1);
      pt[i] = d < pt[i] ? d : pt[i];
      best = __builtin_fmaxf(best, pt[i]);
    }
    // (a member whose lanes own no point at all cannot happen: n >= W * 1024)
    unsigned long long key = ((unsigned long long)__float_as_uint(__builtin_fmaxf(best, 0.f)) << 32) | tiekey;
    if (best < 0.f) key = 0ull;   // padding lane
    key = wave_max_u64(key);
    if (lane == 0) atomicMax(&s_max[par], key);
    lds_barrier();
    key = s_max[par];
    if (t == 0) s_max[1 - par] = 0ull;
    const int tstar = (int)(key & 0xFFFFFu);
    if (t == tstar) {
      // first (lowest k) maximum among this lane's points; then the five granules
      const float vbest = __uint_as_float((unsigned)(key >> 32));
      int sel = 0;
      float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
      for (int i = PW - 1; i >= 0; --i) {
        const bool hit = pt[i] == vbest;
        sel = hit ? t + (w + W * i) * bs : sel;
        sx = hit ? px[i] : sx;
        sy = hit ? py[i] : sy;
        sz = hit ? pz[i] : sz;
      }
      const unsigned lo = ((1023u - rev) << 14) | (unsigned)(16383 - sel);
      unsigned long long *mine = slots + ((size_t)par * W + w) * 8;
      const unsigned long long tag = (unsigned long long)(unsigned)j << 32;
      __hip_atomic_store(mine + 0, tag | (unsigned)(key >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mine + 1, tag | lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mine + 2, tag | __float_as_uint(sx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mine + 3, tag | __float_as_uint(sy), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mine + 4, tag | __float_as_uint(sz), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wave == 0) {
      // all-gather: lane l polls word l % 8 of member l / 8
      const unsigned long long *base = slots + (size_t)par * W * 8;
      unsigned long long x = 0ull;
      bool done = false;
      for (unsigned spins = 0; spins < kFpsSpinLimit; ++spins) {
        if (lane < W * 8 && (lane & 7) < 5)
          x = __hip_atomic_load(base + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        done = __all(!(lane < W * 8 && (lane & 7) < 5) || (unsigned)(x >> 32) == (unsigned)j);
        if (done) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (!done && lane == 0) s_abort = 1;
      const unsigned v = (unsigned)x;
      unsigned long long bestk = 0ull;
      int bw = 0;
#pragma unroll
      for (int q = 0; q < W; ++q) {
        const unsigned long long kq = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)v, q * 8) << 32) |
                                      (unsigned)__builtin_amdgcn_readlane((int)v, q * 8 + 1);
        if (kq > bestk) {
          bestk = kq;
          bw = q;
        }
      }
      if (lane == 0) {
        s_sel[par][0] = __int_as_float(16383 - (int)(bestk & 0x3FFFu));
      }
      // coordinates of the winning member: lanes bw*8+2..4
      const unsigned cx = (unsigned)__builtin_amdgcn_readlane((int)v, bw * 8 + 2 < 64 ? bw * 8 + 2 : 0);
      const unsigned cy = (unsigned)__builtin_amdgcn_readlane((int)v, bw * 8 + 3 < 64 ? bw * 8 + 3 : 0);
      const unsigned cz = (unsigned)__builtin_amdgcn_readlane((int)v, bw * 8 + 4 < 64 ? bw * 8 + 4 : 0);
      if (lane == 0) {
        s_sel[par][1] = __uint_as_float(cx);
        s_sel[par][2] = __uint_as_float(cy);
        s_sel[par][3] = __uint_as_float(cz);
      }
    }
    lds_barrier();
    if (s_abort) break;
    old = __builtin_amdgcn_readfirstlane(__float_as_int(s_sel[par][0]));
    x1 = s_sel[par][1];
    y1 = s_sel[par][2];
    z1 = s_sel[par][3];
    if (w == 0 && t == 0) idxs[j] = old;
  }
  if (s_abort) {   // members were not co-resident (cannot happen under a cooperative launch): fail loudly
    if (w == 0)
      for (int j = t; j < m; j += bs) idxs[j] = -1;
    return;
  }
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int k = t + (w + W * i) * bs;
    if (k < n) temp[k] = pt[i];
  }
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

extern "C" long long mvp_fps_scratch_bytes(int b, int n) {
  if (b < 0 || n < 0) return -1;
  return (long long)b * 16384 * 16;  // one float4 per (padded) point of the largest supported cloud
}

extern "C" int mvp_furthest_point_sampling_sorted(int b, int n, int m, const float *points, float *temp,
                                                  int *idx, void *scratch, long long scratch_bytes,
                                                  void *stream) {
  // the sorted kernel pays off where a lane owns many points; elsewhere the
  // plain kernel is as fast or faster (measured)
  // (measured: 3.22 -> 2.99 ms at (64, 16384 -> 2048); no gain at 8192 points and below)
  if (b <= 0 || m <= 1 || n <= 4096 || n > 16384)
    return mvp_furthest_point_sampling(b, n, m, points, temp, idx, stream);
  if (!points || !temp || !idx || !scratch) return MVP_EBADARG;
  if (scratch_bytes < mvp_fps_scratch_bytes(b, n)) return MVP_EBADARG;
  if ((reinterpret_cast<uintptr_t>(scratch) & 15) != 0) return MVP_EBADARG;
  const int p = (n + 1023) / 1024;
  const int pp = p <= 6 ? 6 : p <= 8 ? 8 : p <= 12 ? 12 : 16;
  float4 *sorted = reinterpret_cast<float4 *>(scratch);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(fps_sort_kernel, dim3(b), dim3(1024), 0, st, n, 1024 * pp, points, sorted);
#define MVP_FPS_SORTED(PP) \
  hipLaunchKernelGGL((fps_sorted_kernel<PP>), dim3(b), dim3(1024), 0, st, n, m, points, sorted, temp, idx)
  if (pp == 6) MVP_FPS_SORTED(6);
  else if (pp == 8) MVP_FPS_SORTED(8);
  else if (pp == 12) MVP_FPS_SORTED(12);
  else MVP_FPS_SORTED(16);
#undef MVP_FPS_SORTED
  return check_launch("mvp_furthest_point_sampling_sorted");
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

extern "C" long long mvp_fps_cluster_scratch_bytes(int b) { return b < 0 ? -1 : (long long)b * (2 * 4 * 8) * 8; }

/* W = 2 or 4 workgroups per cloud; n a multiple of W * 1024, 2048 <= n <= 16384.  scratch:
 * mvp_fps_cluster_scratch_bytes(b) bytes (zeroed here). */
extern "C" int mvp_furthest_point_sampling_cluster(int b, int n, int m, int w, const float *points, float *temp,
                                                   int *idx, void *scratch, long long scratch_bytes, void *stream) {
  if (b < 0 || n <= 0 || m < 0) return MVP_EBADSHAPE;
  if (b == 0 || m == 0) return MVP_OK;
  if ((w != 2 && w != 4) || n % (w * 1024) != 0 || n > 16384) return MVP_EBADSHAPE;
  if (!points || !temp || !idx || !scratch) return MVP_EBADARG;
  if (scratch_bytes < mvp_fps_cluster_scratch_bytes(b)) return MVP_EBADARG;
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(scratch, 0, (size_t)mvp_fps_cluster_scratch_bytes(b), st) != hipSuccess)
    return check_launch("mvp_furthest_point_sampling_cluster");
  int bpad = (b + 7) / 8 * 8;
  const int pw = n / (w * 1024);
  unsigned long long *gr = static_cast<unsigned long long *>(scratch);
  void *args[] = {&b, &bpad, &n, &m, &points, &temp, &idx, &gr};
  const void *fn = nullptr;
#define MVP_FPS_CL(PWV, WV) if (pw == PWV && w == WV) fn = reinterpret_cast<const void *>(fps_cluster_kernel<PWV, WV>)
  MVP_FPS_CL(1, 2); MVP_FPS_CL(2, 2); MVP_FPS_CL(3, 2); MVP_FPS_CL(4, 2); MVP_FPS_CL(5, 2); MVP_FPS_CL(6, 2); MVP_FPS_CL(7, 2); MVP_FPS_CL(8, 2);
  MVP_FPS_CL(1, 4); MVP_FPS_CL(2, 4); MVP_FPS_CL(3, 4); MVP_FPS_CL(4, 4);
#undef MVP_FPS_CL
  if (!fn) return MVP_EBADSHAPE;
  if (hipLaunchCooperativeKernel(fn, dim3(w * bpad), dim3(1024), args, 0, st) != hipSuccess) {
    (void)hipGetLastError();
    return MVP_EBADSHAPE;   // the cluster does not fit this device next to what is running: the caller falls back
  }
  return check_launch("mvp_furthest_point_sampling_cluster");
}
