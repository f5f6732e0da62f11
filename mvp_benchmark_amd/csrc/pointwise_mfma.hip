// Per-point / per-edge linear maps (1x1 convolutions) on the matrix cores: forward,
// data gradient and weight gradient.
//
// the grouped-feature MLPs of the completion networks
// (completion/model_utils.py:26-55 EF_expansion, completion/models/vrcnet.py:21-57
// SA_module, ecg.py:36-65 Dense_conv, pcn.py encoder / decoder convolutions) are
// 1x1 convolutions, i.e. per cloud  Y (Cout x L) = W (Cout x Cin) X (Cin x L) with L
// = points (x neighbours) contiguous.  The reference runs them as cuDNN
// convolutions followed by separate bias / ReLU / residual / max-over-neighbours
// kernels.  Here: LDS-tiled GEMMs on v_mfma_f32_32x32x2_f32 -- float32 in, float32
// accumulate, a k-ordered fmaf chain, no reduced precision (MI355X_MICROARCH.md:
// gfx950 has no xf32 / TF32; the f32 MFMA peak is 157 TFLOP/s) -- with the
// epilogue fused (+ bias, ReLU, + residual, max over groups of consecutive columns)
// and ReLU' applied to the incoming gradient on load in both backward passes.
//
// One compute core, three passes.  An operand tile is staged through LDS in one of
// two images, chosen by how the operand lies in memory, so that every global read
// and every LDS write is a 16-byte access whatever the pass:
//   * "x-contiguous" ([k][x] in memory: the activations X / gY per cloud, and the
//     forward weight read as W^T by the data gradient): LDS image [k][x], an MFMA
//     fragment is one ds_read_b32 per instruction (lanes = 32 consecutive x);
//   * "k-contiguous" ([x][k] in memory: the forward weight, and both operands of the
//     weight gradient, whose reduction dimension is the position): LDS image
//     [x][k + 4], a fragment is one ds_read_b128 per FOUR instructions.
// The k-order inside a group of 8 is permuted the same way for both operands
// (instruction j of a group takes k = j from lanes 0..31 and k = 4 + j from lanes
// 32..63), which is what lets the 16-byte LDS read feed four MFMAs.
// Block = 256 threads = 2 x 2 waves, wave tile 32 TM x 32 TN (TM, TN in {1, 2}), K in
// slabs of BK through double-buffered LDS; the next slab travels global -> registers
// while the current one computes; all fragments of a k-group are read before its
// MFMAs.  Workgroups are numbered so that the ones that share an activation tile run
// back to back on ONE XCD (its L2 serves the re-reads).
#include "common.h"

namespace mvp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMmThreads = 256;
constexpr int kXC = 0, kKC = 1;    // operand images (above)
constexpr int kImgPad = 4;         // floats; keeps rows 16-byte aligned and shifts their banks

template <int BX, int BK, int MODE>
struct PwImg {
  static constexpr int ld = (MODE == kXC ? BX : BK) + kImgPad;
  static constexpr int floats = (MODE == kXC ? BK : BX) * ld;
  static constexpr int vecs = BX * BK / 4 / kMmThreads;     // float4 per thread and slab
  static_assert(BX * BK / 4 % kMmThreads == 0, "slab must divide over the block");
};
template <int V>
struct PwRegs {
  float4 v[V];
};

// One slab of an operand: global -> registers.  g: the operand's matrix (row stride ld floats);
//   kXC: element (k, x) at g[(k0 + k) ld + x0 + x], inside if k0 + k < K and x0 + x < X  (X, ld % 4 == 0)
//   kKC: element (x, k) at g[(x0 + x) ld + k0 + k], inside if x0 + x < X and k0 + k < K
// vec: 16-byte loads are legal (alignment of base and row stride); a float4 along k may straddle K.
// MASK: the mask's values travel in registers of their own and are applied by pw_stash -- a select
// here would make the wave wait for both loads before it starts computing the current slab.
template <int BX, int BK, int MODE, bool MASK>
__device__ __forceinline__ void pw_fetch(PwRegs<BX * BK / 4 / kMmThreads> &r,
                                         PwRegs<MASK ? BX * BK / 4 / kMmThreads : 1> &rm,
                                         const float *__restrict__ g, const float *__restrict__ mask, size_t ld, int x0,
                                         int X, int k0, int K, bool vec, int t) {
  constexpr int V = PwImg<BX, BK, MODE>::vecs;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int q = t + i * kMmThreads;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f), mk = make_float4(1.f, 1.f, 1.f, 1.f);
    if constexpr (MODE == kXC) {
      const int k = q / (BX / 4), x = (q % (BX / 4)) * 4;
      if (k0 + k < K && x0 + x < X) {
        const size_t o = (size_t)(k0 + k) * ld + x0 + x;
        v = *reinterpret_cast<const float4 *>(g + o);
        if constexpr (MASK)
          if (mask) mk = *reinterpret_cast<const float4 *>(mask + o);
      }
    } else {
      const int x = q / (BK / 4), k = (q % (BK / 4)) * 4;
      if (x0 + x < X && k0 + k < K) {
        const size_t o = (size_t)(x0 + x) * ld + k0 + k;
        if (vec && k0 + k + 3 < K) {
          v = *reinterpret_cast<const float4 *>(g + o);
          if constexpr (MASK)
            if (mask) mk = *reinterpret_cast<const float4 *>(mask + o);
        } else {                               // unaligned rows (cin % 4 != 0) or the last, partial group of k
          const float *p = g + o;
          v.x = p[0];
          if (k0 + k + 1 < K) v.y = p[1];
          if (k0 + k + 2 < K) v.z = p[2];
          if (k0 + k + 3 < K) v.w = p[3];
          if constexpr (MASK) {
            if (mask) {
              const float *m = mask + o;
              mk.x = m[0];
              if (k0 + k + 1 < K) mk.y = m[1];
              if (k0 + k + 2 < K) mk.z = m[2];
              if (k0 + k + 3 < K) mk.w = m[3];
            }
          }
        }
      }
    }
    r.v[i] = v;
    if constexpr (MASK) rm.v[i] = mk;
  }
}

// registers -> LDS image (16-byte stores); with MASK, a value counts as 0 where its mask is <= 0; with SELF_RELU
// (round 6; a template parameter: as a run-time flag it cost the forward kernel 16 %) where it is < 0 itself: the operand is relu(.) of what lies in memory -- a pre-activation
// ReLU (vrcnet.py:34,54: conv(relu(x))) costs no pass over x and no second tensor
template <int BX, int BK, int MODE, bool MASK, bool SELF_RELU = false>
__device__ __forceinline__ void pw_stash(float *img, const PwRegs<BX * BK / 4 / kMmThreads> &r,
                                         const PwRegs<MASK ? BX * BK / 4 / kMmThreads : 1> &rm, int t) {
  constexpr int V = PwImg<BX, BK, MODE>::vecs, LD = PwImg<BX, BK, MODE>::ld;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int q = t + i * kMmThreads;
    float4 v = r.v[i];
    if constexpr (SELF_RELU) {                 // (a select, not fmaxf: a NaN stays a NaN, as torch.relu leaves it)
      v.x = v.x < 0.f ? 0.f : v.x;
      v.y = v.y < 0.f ? 0.f : v.y;
      v.z = v.z < 0.f ? 0.f : v.z;
      v.w = v.w < 0.f ? 0.f : v.w;
    }
    if constexpr (MASK) {
      v.x = rm.v[i].x > 0.f ? v.x : 0.f;
      v.y = rm.v[i].y > 0.f ? v.y : 0.f;
      v.z = rm.v[i].z > 0.f ? v.z : 0.f;
      v.w = rm.v[i].w > 0.f ? v.w : 0.f;
    }
    if constexpr (MODE == kXC) {
      const int k = q / (BX / 4), x = (q % (BX / 4)) * 4;
      *reinterpret_cast<float4 *>(img + k * LD + x) = v;
    } else {
      const int x = q / (BK / 4), k = (q % (BK / 4)) * 4;
      *reinterpret_cast<float4 *>(img + x * LD + k) = v;
    }
  }
}

// Fragments of k-group kg (8 values of k) for T MFMA blocks of 32 rows starting at xw:
// v[i][j] is the operand of instruction j: k = 8 kg + j (lanes 0..31) / 8 kg + 4 + j (lanes 32..63).
template <int T, int BX, int BK, int MODE>
__device__ __forceinline__ void pw_frag(float (&v)[T][4], const float *img, int xw, int kg, int lrow, int lk) {
  constexpr int LD = PwImg<BX, BK, MODE>::ld;
#pragma unroll
  for (int i = 0; i < T; ++i) {
    if constexpr (MODE == kXC) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[i][j] = img[(kg * 8 + j + 4 * lk) * LD + xw + i * 32 + lrow];
    } else {
      const float4 q = *reinterpret_cast<const float4 *>(img + (xw + i * 32 + lrow) * LD + kg * 8 + 4 * lk);
      v[i][0] = q.x;
      v[i][1] = q.y;
      v[i][2] = q.z;
      v[i][3] = q.w;
    }
  }
}

// k-group of a slab after whose instructions the next slab is written to LDS.  The weight gradient (slabs of 32,
// two workgroups per CU) writes it before the slab's last group, so that the stores' latency and the wait for the
// global loads sit under that group's 16 matrix instructions: 87 -> 98 TFLOP/s at (64, 512 -> 1024, 2048).  The
// forward / data-gradient kernel (slabs of 16, three workgroups per CU) measured 1-3 % slower that way.
constexpr int kPwStashAt(int groups, bool early) { return early && groups >= 2 ? groups - 2 : groups - 1; }

template <int TM, int TN>
__device__ __forceinline__ void pw_mma(f32x16 (&acc)[TM][TN], const float (&a)[TM][4], const float (&b)[TN][4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int n = 0; n < TN; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], b[n][j], acc[i][n], 0, 0, 0);
}

// Unsigned max with one DPP-modified operand (lanes a row mask leaves unwritten combine with 0, the identity).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned pw_dpp_umax(unsigned v) {
  const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
  return o > v ? o : v;
}

// keys (see pointwise_mfma_kernel, ROWMAX) -> val[i] = the row's maximum, idx[i] = the first position that attains it
__global__ void pointwise_rowmax_unpack_kernel(long long total, const unsigned long long *__restrict__ keys,
                                               float *__restrict__ val, int *__restrict__ idx) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const unsigned long long k = keys[i];
  const unsigned o = (unsigned)(k >> 32);
  val[i] = __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
  idx[i] = (int)~(unsigned)k;
}

// Workgroup number -> work item so that consecutive items run on ONE XCD (workgroups are dealt to the 8 XCDs
// round-robin by their linear id): XCD x takes items [x per, (x + 1) per).
__device__ __forceinline__ long long pw_work_item(long long total) {
  const long long per = (total + 7) / 8;
  return (long long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
}

// ------------------------------------------------------------------- forward / data gradient
// Per cloud  Y (M x N) = A (M x K) X (K x N): A = the weight, (M, K) row-major [AMODE kKC: forward] or (K, M)
// row-major [kXC: the data gradient multiplies by W^T]; X, xmask (K x N), Y per cloud.
// ROWMAX (round 5): the epilogue keeps only the maximum of every output row over ALL positions of its cloud, and where it
// was attained -- the PointNet stage of the completion networks (conv -> max over the points: pcn.py:25-31 conv4, the
// relational encoder's conv5) writes a (B, Cout, N) tensor only to reduce it in the next kernel (0.5 GB each way at
// 64 x 1024 x 2048).  `y` then is a (nb, M) array of 64-bit keys {order-preserving bits of the value | ~column}, zeroed by the
// caller; a tile's rows join by one 64-bit atomic max per row and half-wave (the lanes that hold the half-wave's maximum
// issue it: ties go to the smallest column), pointwise_rowmax_unpack_kernel turns the keys into values and positions.
// (waves_per_eu: the residual / two-output epilogue of round 6 would otherwise lift the allocation from 168 to 188 VGPRs
// -- two workgroups per CU instead of three for EVERY call, 0.4 ms of a VRCNet step)
template <int TM, int TN, int BK, int AMODE, bool MASKED, bool ROWMAX = false, bool XRELU = false>
__global__ __launch_bounds__(kMmThreads) __attribute__((amdgpu_waves_per_eu(TM == 2 ? 3 : 5))) void pointwise_mfma_kernel(
    int M, int N, int K, int nb, const float *__restrict__ a, int a_ld, int a_vec, const float *__restrict__ x,
    const float *__restrict__ xmask, const float *__restrict__ bias, const float *__restrict__ residual, int flags,
    int group, float *__restrict__ y, int bias_bs = 0, int m_split = 0, float *__restrict__ y2 = nullptr) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  using IA = PwImg<BM, BK, AMODE>;
  using IB = PwImg<BN, BK, kXC>;
  // flags (mvpops.h MVP_PW_*): ReLU before the residual | ReLU after it | the residual is a MASK (the value counts as 0
  // where it is <= 0: the data gradient of conv(relu(x))) | ReLU of x on load
  const bool relu = (flags & 1) != 0, relu_after = (flags & 2) != 0, res_is_mask = (flags & 4) != 0;
  static_assert(!(XRELU && MASKED), "ReLU of x on load and a mask on x are not combined");
  __shared__ __attribute__((aligned(16))) float As[2][IA::floats];
  __shared__ __attribute__((aligned(16))) float Bs[2][IB::floats];
  __shared__ float sbias[BM];

  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const long long total = (long long)tiles_m * tiles_n * nb;
  const long long work = pw_work_item(total);
  if (work >= total) return;
  const int m0 = (int)(work % tiles_m) * BM;               // the M tiles of one activation tile are neighbours
  const int n0 = (int)((work / tiles_m) % tiles_n) * BN;
  const int cloud = (int)(work / ((long long)tiles_m * tiles_n));

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1, lrow = lane & 31, lk = lane >> 5;
  const float *xb = x + (size_t)cloud * K * N;
  const float *mb = xmask ? xmask + (size_t)cloud * K * N : nullptr;
  const size_t lda = a_ld;

  if (t < BM) sbias[t] = (bias && m0 + t < M) ? bias[(size_t)cloud * bias_bs + m0 + t] : 0.f;   // bias_bs = M: a bias per cloud

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  PwRegs<IA::vecs> ra;
  PwRegs<IB::vecs> rb;
  PwRegs<1> rnone;
  PwRegs<MASKED ? IB::vecs : 1> rbm;
  const int nk = (K + BK - 1) / BK;
  pw_fetch<BM, BK, AMODE, false>(ra, rnone, a, nullptr, lda, m0, M, 0, K, a_vec != 0, t);
  pw_fetch<BN, BK, kXC, MASKED>(rb, rbm, xb, mb, N, n0, N, 0, K, true, t);
  pw_stash<BM, BK, AMODE, false>(As[0], ra, rnone, t);
  pw_stash<BN, BK, kXC, MASKED, XRELU>(Bs[0], rb, rbm, t);
  __syncthreads();
  for (int s = 0; s < nk; ++s) {
    const int buf = s & 1;
    if (s + 1 < nk) {                                   // in flight while this slab computes
      pw_fetch<BM, BK, AMODE, false>(ra, rnone, a, nullptr, lda, m0, M, (s + 1) * BK, K, a_vec != 0, t);
      pw_fetch<BN, BK, kXC, MASKED>(rb, rbm, xb, mb, N, n0, N, (s + 1) * BK, K, true, t);
    }
#pragma unroll
    for (int kg = 0; kg < BK / 8; ++kg) {
      float av[TM][4], bv[TN][4];
      pw_frag<TM, BM, BK, AMODE>(av, As[buf], wm * 32 * TM, kg, lrow, lk);
      pw_frag<TN, BN, BK, kXC>(bv, Bs[buf], wn * 32 * TN, kg, lrow, lk);
      pw_mma<TM, TN>(acc, av, bv);
      if (kg == kPwStashAt(BK / 8, false) && s + 1 < nk) {
        // the next slab goes into the other buffer (nobody reads it in this step) while the matrix
        // core works through the instructions issued so far: the stores' latency is covered
        pw_stash<BM, BK, AMODE, false>(As[buf ^ 1], ra, rnone, t);
        pw_stash<BN, BK, kXC, MASKED, XRELU>(Bs[buf ^ 1], rb, rbm, t);
      }
    }
    if (s + 1 < nk) __syncthreads();
  }

  // ---- epilogue.  C/D layout of 32x32: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  if constexpr (ROWMAX) {
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(y) + (size_t)cloud * M;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        const float bi = sbias[lr];
        unsigned hi = 0u, lo = 0u;            // this lane's best of its TN columns: value bits (0: none), ~column
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = n0 + wn * 32 * TN + j * 32 + lrow;
          float v = acc[i][j][r] + bi;
          if (relu) v = __builtin_fmaxf(v, 0.f);
          // order-preserving key, > 0; -0 counts as +0 (the first position of a tie wins, as torch.max reports) and a NaN
          // of either sign takes the top key (torch.max propagates NaN; unpacked as a NaN again)
          const unsigned b = __float_as_uint(v + 0.f);
          const unsigned o = col < N ? (v != v ? 0xFFFFFFFFu : (b & 0x80000000u) ? ~b : (b | 0x80000000u)) : 0u;
          if (o > hi) { hi = o; lo = ~(unsigned)col; }                                         // (strict: the smaller column stays)
        }
        // maximum over the 32 lanes that share this row (lanes 0..31 / 32..63): DPP inside the rows of 16, row_bcast15
        // into rows 1 and 3, whose last lanes then hold it
        unsigned m = hi;
        m = pw_dpp_umax<0xB1, 0xF>(m);    // quad_perm [1,0,3,2]
        m = pw_dpp_umax<0x4E, 0xF>(m);    // quad_perm [2,3,0,1]
        m = pw_dpp_umax<0x141, 0xF>(m);   // row_half_mirror
        m = pw_dpp_umax<0x140, 0xF>(m);   // row_mirror
        m = pw_dpp_umax<0x142, 0xA>(m);   // row_bcast15 -> rows 1, 3
        const unsigned m0h = (unsigned)__builtin_amdgcn_readlane((int)m, 31), m1h = (unsigned)__builtin_amdgcn_readlane((int)m, 63);
        const unsigned mh = lk ? m1h : m0h;
        if (hi != 0u && hi == mh && m0 + lr < M)
          atomicMax(keys + (m0 + lr), ((unsigned long long)hi << 32) | (unsigned long long)lo);
      }
    }
    return;
  }
  const int len_out = N / group;
  const float *rbse = residual ? residual + (size_t)cloud * M * len_out : nullptr;
  const bool split = m_split > 0 && m_split < M;
  if (group == 1 && !rbse && !split) {                   // the common case: straight-line
    float *yb = y + (size_t)cloud * M * N;
    const bool relu_any = relu || relu_after;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float bv[16];                                      // the rows' biases first: independent LDS reads, one wait
#pragma unroll
      for (int r = 0; r < 16; ++r) bv[r] = sbias[wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * 32 * TN + j * 32 + lrow;
        float *yc = yb + (size_t)(m0 + wm * 32 * TM + i * 32 + 4 * lk) * N + col;
        const int rows_left = M - (m0 + wm * 32 * TM + i * 32 + 4 * lk);   // rows of this lane that exist
        if (col < N) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            float v = acc[i][j][r] + bv[r];
            v = relu_any ? __builtin_fmaxf(v, 0.f) : v;
            if (dr < rows_left) yc[(size_t)dr * N] = v;
          }
        }
      }
    }
    return;
  }
  if (group == 1) {
    // round 6: + residual (or the residual as a mask) and a second ReLU, or two outputs -- straight-line like the common
    // case, a block's 16 residual values fetched before the first is used.  Two outputs (convolutions of ONE input stacked
    // along the output channels -- a residual unit's conv1 / conv_res, vrcnet.py:160-172 -- hand each consumer a contiguous
    // tensor, no torch.split views and no copy): rows below m_split go to y (m_split rows per cloud), the others to y2
    // (M - m_split rows); m_split % 32 == 0, so a 32-row MFMA block lies on one side.
    const int ms = split ? m_split : M;
    float *y1b = y + (size_t)cloud * ms * N;
    float *y2b = split ? y2 + ((size_t)cloud * (M - ms) - (size_t)ms) * N : y1b;     // indexed by the global row
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float bv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) bv[r] = sbias[wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk];
      const int blk = m0 + wm * 32 * TM + i * 32;        // first row of this 32-row block
      float *yb = blk < ms ? y1b : y2b;
      const int rows_left = (blk < ms ? ms : M) - (blk + 4 * lk);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * 32 * TN + j * 32 + lrow;
        const size_t o0 = (size_t)(blk + 4 * lk) * N + col;
        if (col < N) {
          float rv[16];
          if (rbse) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int dr = (r & 3) + 8 * (r >> 2);
              rv[r] = dr < rows_left ? rbse[o0 + (size_t)dr * N] : 0.f;
            }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            float v = acc[i][j][r] + bv[r];
            v = relu ? __builtin_fmaxf(v, 0.f) : v;
            if (rbse) v = res_is_mask ? (rv[r] > 0.f ? v : 0.f) : v + rv[r];
            v = relu_after ? __builtin_fmaxf(v, 0.f) : v;
            if (dr < rows_left) yb[o0 + (size_t)dr * N] = v;
          }
        }
      }
    }
    return;
  }
  float *yb = y + (size_t)cloud * M * len_out;             // groups of columns reduced with max (no split: checked by the host)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * 32 * TN + j * 32 + lrow;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int row = m0 + lr;
        float v = acc[i][j][r] + sbias[lr];
        if (relu) v = __builtin_fmaxf(v, 0.f);
        if (col >= N) v = -__builtin_inff();              // (N % group == 0: a group is inside or outside)
        for (int off = 1; off < group; off <<= 1) v = __builtin_fmaxf(v, __shfl_xor(v, off, 64));
        if (row < M && col < N && (lrow & (group - 1)) == 0) {
          const size_t o = (size_t)row * len_out + col / group;
          if (rbse) {
            const float rv = rbse[o];
            v = res_is_mask ? (rv > 0.f ? v : 0.f) : v + rv;
          }
          if (relu_after) v = __builtin_fmaxf(v, 0.f);
          yb[o] = v;
        }
      }
    }
}

// ---------------------------------------------------------------- weight gradient
// gw[co][ci] = sum_b sum_l g[b][co][l] x[b][ci][l]  (g = grad_out, optionally masked by ReLU'),
// gb[co] = sum_b sum_l g[b][co][l].  M = cout, N = cin, K = the b * len positions: a huge reduction with a
// small output.  The positions -- slabs of BK per cloud, numbered across the clouds -- are dealt out in
// contiguous runs to `splits` workgroups per output tile, as many as fit the chip at once (one wave of
// workgroups, equal work, no tail); each writes its partial tile, a second kernel adds the partials in a fixed
// order (no float atomics: reproducible).  The bias gradient is the row sum of the staged g tile, taken from
// LDS by the workgroups of the first column of tiles.  Both operands are k-contiguous.
template <int TM, int TN, int BK, bool MASKED, bool XRELU = false>
__global__ __launch_bounds__(kMmThreads) void pointwise_wgrad_mfma_kernel(
    int nb, int cin, int cout, int len, int slabs_per_cloud, int slabs_per_split, int splits,
    const float *__restrict__ x, const float *__restrict__ g, const float *__restrict__ gmask, int with_bias,
    float *__restrict__ partial, float *__restrict__ pbias) {   // XRELU: the layer was conv(relu(x)): x counts as relu(x)
  constexpr int BM = 64 * TM, BN = 64 * TN;
  using IA = PwImg<BM, BK, kKC>;
  using IB = PwImg<BN, BK, kKC>;
  __shared__ __attribute__((aligned(16))) float As[2][IA::floats];
  __shared__ __attribute__((aligned(16))) float Bs[2][IB::floats];

  const int tiles_m = (cout + BM - 1) / BM, tiles_n = (cin + BN - 1) / BN, tiles = tiles_m * tiles_n;
  const long long total = (long long)tiles * splits;
  const long long work = pw_work_item(total);
  if (work >= total) return;
  const int tile = (int)(work % tiles), split = (int)(work / tiles);   // the tiles of one run of positions are neighbours
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int s_begin = split * slabs_per_split;
  const int s_end = min(s_begin + slabs_per_split, nb * slabs_per_cloud);

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1, lrow = lane & 31, lk = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const bool do_bias = with_bias && n0 == 0;
  constexpr int kParts = kMmThreads / BM;                   // threads per row of the g tile
  float bsum = 0.f;

  PwRegs<IA::vecs> ra;
  PwRegs<IB::vecs> rb;
  PwRegs<1> rnone;
  PwRegs<MASKED ? IA::vecs : 1> ram;
  auto fetch = [&](int gs) {
    const int cloud = gs / slabs_per_cloud, l0 = (gs - cloud * slabs_per_cloud) * BK;
    const size_t og = (size_t)cloud * cout * len, ox = (size_t)cloud * cin * len;
    pw_fetch<BM, BK, kKC, MASKED>(ra, ram, g + og, MASKED ? gmask + og : nullptr, len, m0, cout, l0, len, true, t);
    pw_fetch<BN, BK, kKC, false>(rb, rnone, x + ox, nullptr, len, n0, cin, l0, len, true, t);
  };
  if (s_begin < s_end) {
    fetch(s_begin);
    pw_stash<BM, BK, kKC, MASKED>(As[0], ra, ram, t);
    pw_stash<BN, BK, kKC, false, XRELU>(Bs[0], rb, rnone, t);
  }
  __syncthreads();
  for (int s = s_begin; s < s_end; ++s) {
    const int buf = (s - s_begin) & 1;
    if (s + 1 < s_end) fetch(s + 1);
#pragma unroll
    for (int kg = 0; kg < BK / 8; ++kg) {
      float av[TM][4], bv[TN][4];
      pw_frag<TM, BM, BK, kKC>(av, As[buf], wm * 32 * TM, kg, lrow, lk);
      pw_frag<TN, BN, BK, kKC>(bv, Bs[buf], wn * 32 * TN, kg, lrow, lk);
      pw_mma<TM, TN>(acc, av, bv);
      if (kg == kPwStashAt(BK / 8, true) && s + 1 < s_end) {
        pw_stash<BM, BK, kKC, MASKED>(As[buf ^ 1], ra, ram, t);
        pw_stash<BN, BK, kKC, false, XRELU>(Bs[buf ^ 1], rb, rnone, t);
      }
    }
    if (do_bias) {
      const float *row = As[buf] + (t / kParts) * IA::ld + (t % kParts) * (BK / kParts);
#pragma unroll
      for (int k = 0; k < BK / kParts; k += 4) {
        const float4 q = *reinterpret_cast<const float4 *>(row + k);
        bsum += (q.x + q.y) + (q.z + q.w);
      }
    }
    if (s + 1 < s_end) __syncthreads();
  }
  // partial[split][co][ci]
  float *pp = partial + (size_t)split * cout * cin;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * 32 * TN + j * 32 + lrow;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row < cout && col < cin) pp[(size_t)row * cin + col] = acc[i][j][r];
      }
    }
  if (do_bias) {
#pragma unroll
    for (int off = 1; off < kParts; off <<= 1) bsum += __shfl_xor(bsum, off, 64);
    const int row = m0 + t / kParts;
    if (t % kParts == 0 && row < cout) pbias[(size_t)split * cout + row] = bsum;
  }
}

// gw[co][ci] (and gb[co]) = sum over the splits in a fixed order: a block owns 64 outputs, its wave g adds the
// splits g, g + 4, g + 8, ... (four running sums, 16 loads in flight per output), the four waves' sums are
// combined through LDS.  (One thread per output walking all the splits was a chain of dependent-latency
// round trips: 18 us per call, 0.69 ms per VRCNet step.)
__global__ __launch_bounds__(256) void pointwise_wgrad_reduce_kernel(int cin, int cout, int splits,
                                                                     const float *__restrict__ partial,
                                                                     const float *__restrict__ pbias,
                                                                     float *__restrict__ gw, float *__restrict__ gb) {
  __shared__ float part[4][64];
  const long long nw = (long long)cout * cin, total = nw + (gb ? cout : 0);
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  long long e = (long long)blockIdx.x * 64 + lane;
  const bool valid = e < total;
  const float *src = partial;
  long long stride = nw;
  float *dst = gw;
  if (e >= nw) {
    e -= nw;
    src = pbias;
    stride = cout;
    dst = gb;
  }
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (valid) {
    int sp = g;
    for (; sp + 12 < splits; sp += 16) {
      s0 += src[(size_t)sp * stride + e];
      s1 += src[(size_t)(sp + 4) * stride + e];
      s2 += src[(size_t)(sp + 8) * stride + e];
      s3 += src[(size_t)(sp + 12) * stride + e];
    }
    for (; sp < splits; sp += 4) s0 += src[(size_t)sp * stride + e];
  }
  part[g][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && valid) dst[e] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

#ifndef MVP_WG_BK
#define MVP_WG_BK 32
#endif
constexpr int kWgBK = MVP_WG_BK;
// workgroups of the weight-gradient kernel resident at once: BK 32 -> 74 KB of LDS, 240 VGPRs: 2 per CU; BK 16: 3 per CU
constexpr int kWgSlots = kWgBK == 32 ? 512 : 768;

// runs of position slabs: splits x tiles ~ one wave of workgroups
static void wgrad_plan(int b, int cin, int cout, int len, int &slabs_per_cloud, int &slabs_per_split, int &splits) {
  const int bm = cout > 64 ? 128 : 64, bn = cin > 64 ? 128 : 64;
  const long long tiles = (long long)((cout + bm - 1) / bm) * ((cin + bn - 1) / bn);
  slabs_per_cloud = (len + kWgBK - 1) / kWgBK;
  const long long slabs = (long long)b * slabs_per_cloud;
  // (64 x 64 tiles: 37 KB of LDS, 100 VGPRs -- four workgroups per CU)
  long long want = (bm == 64 && bn == 64 ? 2 * kWgSlots : kWgSlots) / tiles;
  if (want < 1) want = 1;
  if (want > slabs) want = slabs;
  slabs_per_split = (int)((slabs + want - 1) / want);
  if (slabs_per_split < 4 && slabs >= 4) slabs_per_split = 4;     // a workgroup's prologue wants a few slabs behind it
  splits = (int)((slabs + slabs_per_split - 1) / slabs_per_split);
}

}  // namespace mvp

using namespace mvp;

extern "C" long long mvp_pointwise_wgrad_mfma_scratch_bytes(int b, int cin, int cout, int len, int with_bias) {
  if (b <= 0 || cin <= 0 || cout <= 0 || len <= 0 || (len & 3) != 0) return 0;
  if ((long long)b * ((len + kWgBK - 1) / kWgBK) > 2147483647LL) return 0;
  int spc, sps, splits;
  wgrad_plan(b, cin, cout, len, spc, sps, splits);
  return (long long)splits * cout * (cin + (with_bias ? 1 : 0)) * 4;
}

extern "C" int mvp_pointwise_wgrad_mfma(int b, int cin, int cout, int len, const float *x, const float *gy,
                                        const float *gymask, float *gw, float *gb, void *scratch,
                                        long long scratch_bytes, void *stream) {
  return mvp_pointwise_wgrad_mfma_ex(b, cin, cout, len, x, 0, gy, gymask, gw, gb, scratch, scratch_bytes, stream);
}

extern "C" int mvp_pointwise_wgrad_mfma_ex(int b, int cin, int cout, int len, const float *x, int x_relu, const float *gy,
                                           const float *gymask, float *gw, float *gb, void *scratch,
                                           long long scratch_bytes, void *stream) {
  const int with_bias = gb != nullptr;
  const long long need = mvp_pointwise_wgrad_mfma_scratch_bytes(b, cin, cout, len, with_bias);
  if (need == 0) return MVP_EBADSHAPE;
  if (!x || !gy || !gw || !scratch || scratch_bytes < need) return MVP_EBADARG;
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(gymask)) & 15) != 0)
    return MVP_EBADARG;
  int spc, sps, splits;
  wgrad_plan(b, cin, cout, len, spc, sps, splits);
  hipStream_t st = as_stream(stream);
  float *partial = static_cast<float *>(scratch);
  float *pbias = partial + (size_t)splits * cout * cin;
  const int bm = cout > 64 ? 128 : 64, bn = cin > 64 ? 128 : 64;
  const long long total = (long long)((cout + bm - 1) / bm) * ((cin + bn - 1) / bn) * splits;
  const long long nwg = (total + 7) / 8 * 8;
  if (nwg > 2147483647LL) return MVP_EBADSHAPE;
#define MVP_WG_(TM, TN, MK, XR)                                                                                             \
  hipLaunchKernelGGL((pointwise_wgrad_mfma_kernel<TM, TN, kWgBK, MK, XR>), dim3((unsigned)nwg), dim3(kMmThreads), 0, st, b, \
                     cin, cout, len, spc, sps, splits, x, gy, gymask, with_bias, partial, pbias)
#define MVP_WG(TM, TN)                                \
  do {                                                \
    if (gymask && x_relu) MVP_WG_(TM, TN, true, true);   \
    else if (gymask) MVP_WG_(TM, TN, true, false);       \
    else if (x_relu) MVP_WG_(TM, TN, false, true);       \
    else MVP_WG_(TM, TN, false, false);                  \
  } while (0)
  if (bm == 128) {
    if (bn == 128) MVP_WG(2, 2);
    else MVP_WG(2, 1);
  } else {
    if (bn == 128) MVP_WG(1, 2);
    else MVP_WG(1, 1);
  }
#undef MVP_WG
#undef MVP_WG_
  const long long elems = (long long)cout * cin + (with_bias ? cout : 0);
  hipLaunchKernelGGL(pointwise_wgrad_reduce_kernel, dim3((unsigned)((elems + 63) / 64)), dim3(256), 0, st, cin, cout,
                     splits, partial, pbias, gw, gb);
  return check_launch("mvp_pointwise_wgrad_mfma");
}


extern "C" int mvp_pointwise_mfma(int b, int cin, int cout, int len, const float *x, const float *xmask,
                                  const float *w, int ldw, int w_kmajor, const float *bias, const float *residual,
                                  int relu, int group, float *y, void *stream) {
  return mvp_pointwise_mfma_ex(b, cin, cout, len, x, xmask, w, ldw, w_kmajor, bias, 0, residual, relu ? MVP_PW_RELU : 0,
                               group, y, 0, nullptr, stream);
}

extern "C" int mvp_pointwise_mfma_ex(int b, int cin, int cout, int len, const float *x, const float *xmask,
                                     const float *w, int ldw, int w_kmajor, const float *bias, int bias_per_cloud,
                                     const float *residual, int flags, int group, float *y, int m_split, float *y2,
                                     void *stream) {
  if (b < 0 || cin <= 0 || cout <= 0 || len < 0) return MVP_EBADSHAPE;
  if ((flags & ~(MVP_PW_RELU | MVP_PW_RELU_AFTER | MVP_PW_RES_IS_MASK | MVP_PW_X_RELU)) != 0) return MVP_EBADARG;
  if ((flags & MVP_PW_RES_IS_MASK) && !residual) return MVP_EBADARG;
  if ((flags & MVP_PW_X_RELU) && xmask) return MVP_EBADARG;
  if (m_split != 0 && (m_split < 0 || m_split >= cout || (m_split & 31) != 0 || !y2 || residual || group != 1)) return MVP_EBADARG;
  if (bias_per_cloud && !bias) return MVP_EBADARG;
  const int bias_bs = bias_per_cloud ? cout : 0;
  if (group < 1 || group > 32 || (group & (group - 1)) != 0) return MVP_EBADSHAPE;
  if ((len & 3) != 0 || len % group != 0 || b > 65535) return MVP_EBADSHAPE;
  if (b == 0 || len == 0) return MVP_OK;
  if (!x || !w || !y) return MVP_EBADARG;
  if ((reinterpret_cast<uintptr_t>(x) & 15) != 0 || (reinterpret_cast<uintptr_t>(xmask) & 15) != 0) return MVP_EBADARG;
  if (ldw == 0) ldw = w_kmajor ? cout : cin;                    // dense rows
  if (ldw < (w_kmajor ? cout : cin)) return MVP_EBADARG;
  if (w_kmajor && ((cout & 3) != 0 || (ldw & 3) != 0 || (reinterpret_cast<uintptr_t>(w) & 15) != 0)) return MVP_EBADARG;
  if (!w_kmajor && (ldw & 3) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) != 0) return MVP_EBADARG;
  hipStream_t st = as_stream(stream);
  const bool big = cout > 64;
  const int bm = big ? 128 : 64;
  const long long total = (long long)((cout + bm - 1) / bm) * ((len + 127) / 128) * b;
  const long long nwg = (total + 7) / 8 * 8;
  if (nwg > 2147483647LL) return MVP_EBADSHAPE;
  const int a_vec = w_kmajor ? 1 : ((ldw & 3) == 0);            // 16-byte loads along k need aligned rows
#define MVP_MM_(TM, MODE, MK, XR)                                                                                       \
  hipLaunchKernelGGL((pointwise_mfma_kernel<TM, 2, 16, MODE, MK, false, XR>), dim3((unsigned)nwg), dim3(kMmThreads), 0, st, cout, \
                     len, cin, b, w, ldw, a_vec, x, xmask, bias, residual, flags, group, y, bias_bs, m_split, y2)
#define MVP_MM(TM, MODE)                                             \
  do {                                                               \
    if (xmask) MVP_MM_(TM, MODE, true, false);                       \
    else if (flags & MVP_PW_X_RELU) MVP_MM_(TM, MODE, false, true);  \
    else MVP_MM_(TM, MODE, false, false);                            \
  } while (0)
  if (big) {
    if (w_kmajor) MVP_MM(2, kXC);
    else MVP_MM(2, kKC);
  } else {
    if (w_kmajor) MVP_MM(1, kXC);
    else MVP_MM(1, kKC);
  }
#undef MVP_MM
#undef MVP_MM_
  return check_launch("mvp_pointwise_mfma");
}

// (W x + bias).max over the positions of every cloud, fused: val (b, cout), idx (b, cout) = the first position that
// attains it; keys: b * cout * 8 bytes of scratch (contents irrelevant).  Same arithmetic as mvp_pointwise_mfma (the same
// k-ordered fmaf chain per output), so val == mvp_pointwise_mfma's output reduced with max, bit for bit.
extern "C" int mvp_pointwise_mfma_max(int b, int cin, int cout, int len, const float *x, const float *w, int ldw,
                                      const float *bias, int relu, float *val, int *idx, void *keys, long long keys_bytes,
                                      void *stream) {
  if (b < 0 || cin <= 0 || cout <= 0 || len <= 0) return MVP_EBADSHAPE;
  if ((len & 3) != 0 || b > 65535) return MVP_EBADSHAPE;
  if (b == 0) return MVP_OK;
  if (!x || !w || !val || !idx || !keys) return MVP_EBADARG;
  if (keys_bytes < (long long)b * cout * 8 || (reinterpret_cast<uintptr_t>(keys) & 7) != 0) return MVP_EBADARG;
  if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) return MVP_EBADARG;
  if (ldw == 0) ldw = cin;
  if (ldw < cin) return MVP_EBADARG;
  if ((ldw & 3) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) != 0) return MVP_EBADARG;
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(keys, 0, (size_t)b * cout * 8, st) != hipSuccess) return MVP_ELAUNCH;
  const bool big = cout > 64;
  const int bm = big ? 128 : 64;
  const long long total = (long long)((cout + bm - 1) / bm) * ((len + 127) / 128) * b;
  const long long nwg = (total + 7) / 8 * 8;
  if (nwg > 2147483647LL) return MVP_EBADSHAPE;
  const int a_vec = (ldw & 3) == 0;
  const float *none = nullptr;
  float *y = reinterpret_cast<float *>(keys);
  const int group = 1;
  if (big)
    hipLaunchKernelGGL((pointwise_mfma_kernel<2, 2, 16, kKC, false, true>), dim3((unsigned)nwg), dim3(kMmThreads), 0, st, cout, len,
                       cin, b, w, ldw, a_vec, x, none, bias, none, relu, group, y);
  else
    hipLaunchKernelGGL((pointwise_mfma_kernel<1, 2, 16, kKC, false, true>), dim3((unsigned)nwg), dim3(kMmThreads), 0, st, cout, len,
                       cin, b, w, ldw, a_vec, x, none, bias, none, relu, group, y);
  const long long rows = (long long)b * cout;
  hipLaunchKernelGGL(pointwise_rowmax_unpack_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, rows,
                     reinterpret_cast<const unsigned long long *>(keys), val, idx);
  return check_launch("mvp_pointwise_mfma_max");
}
