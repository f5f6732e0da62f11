// Shared device helpers of the EMD auction kernels (emd.hip: the clustered first
// kernel; emd_lean.hip: the kernel of the one-bidder-per-wave rounds).  See emd.hip for the
// design notes and the reference citations.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace mvp {

constexpr int kEmdThreads = 1024;
constexpr int kEmdWaves = kEmdThreads / kWave;
constexpr int kIndexReserve = 1728;  // ints of per-cloud scratch the grid of rounds 1-5 kept its cell offsets in (the size formula of the scratch stays)
constexpr int kBidCache = 1024;  // list positions whose bid is cached in LDS
constexpr int kRecCap = 512;     // list positions whose person record is cached in LDS
#ifndef MVP_EMD_ROWMIN
#define MVP_EMD_ROWMIN 192   // (96 with the 12^3 grid of rounds 1-5: below)
#endif
constexpr int kRowModeMin = MVP_EMD_ROWMIN;  // bidders per round above which 4 bidders share a wave
constexpr int kRowListCap = 128; // per-row (bidder) list of surviving leaves, flushed when full
constexpr int kMaxCluster = 8;   // workgroups per cloud (W)
constexpr int kChgCap = 2048;    // refreshed price bounds a workgroup can broadcast per round
#ifndef MVP_EMD_SOLO
#define MVP_EMD_SOLO 16
#endif
// unassigned persons below which one workgroup finishes the auction alone (32 / 64 / 128 with the
// four-bidders-per-wave schedule were tried in round 2: 5 % / 23 % / 57 % slower at the headline shape)
constexpr int kSoloMax = MVP_EMD_SOLO;
#ifndef MVP_EMD_FEW
#define MVP_EMD_FEW 1   // a collapsed cluster's rounds with few bidders per wave: emd_lean_round_few.inc (0: the plain rounds, A/B builds)
#endif
#ifndef MVP_EMD_FEWPOS
#define MVP_EMD_FEWPOS 1   // ... list positions per wave there: the lean kernels' cluster collapses to member 0 at 16 x this many persons
#endif
constexpr int kFewPos = MVP_EMD_FEWPOS;
constexpr int kFewMax = kFewPos * kEmdWaves;
constexpr unsigned kSpinLimit = 1u << 24;  // bound of every cluster wait (tens of seconds), then abort
// The lean kernel (emd_lean.hip) takes a cloud over once no workgroup has more than kRowModeMin
// bidders, at most kLeanCap persons are unassigned (their number never grows, so every later
// list fits the LDS record cache) and at least kLeanMinRounds rounds are left.
// Re-tuned with the leaf index (round 6, same-box sweep of six builds, profiles/r6c_emd_handover_sweep.txt): the one-bidder-
// per-wave searches got 14 % cheaper and the four-per-wave ones 14 % dearer, so the lean kernel takes over EARLIER --
// kRowModeMin x kLeanCap = 96 x 384 (rounds 2-5: "64..160 x 384 / 512 all within 0.2 ms") -> 192 x 512 (the record cache's
// size): headline EMD 29.5 -> 29.0 ms, 256 x 512 the same, 64 x 384 +1.6; every other shape neutral or better.
#ifndef MVP_EMD_LEANCAP
#define MVP_EMD_LEANCAP 512
#endif
constexpr int kLeanCap = MVP_EMD_LEANCAP;
constexpr int kLeanMinRounds = 64;
static_assert(kLeanCap <= kRecCap, "the lean kernel keeps every list entry's record in LDS");
// The resident kernel (emd_resident.hip) takes a cloud of at most kResMaxN points over once at most
// `res_cap` <= kResList persons are unassigned and at least kResMinRounds rounds are left: the whole
// auction state then lives in one workgroup's LDS.
// Gathered-bid rounds of the lean kernels (emd_lean.hip): once a cloud has at most kGCap unassigned persons its bids
// travel as tagged 16-byte granules, every member of the cluster settles every bid itself and the round has ONE
// cluster-wide wait instead of a bid atomic, two all-gathers and Assign's dependent loads.
constexpr int kGCap = 256;      // bids of a cloud per round
constexpr int kGMaxN = 16384;   // 14-bit slots / person indices in a granule
constexpr int kGStride = 2 * kGCap + 8;   // 8-byte words of one parity's bid area: the bids, then a heartbeat per member
constexpr size_t kEmdBidBytes = 2 * (size_t)kGStride * 8;   // per cloud, both parities
constexpr int kResMaxN = 4096;
constexpr int kResList = 64;
constexpr int kResMinRounds = 32;

// Filter slack.  An object is skipped only if
//   s > fl(tq*tq),  tq = fl(fl(fl(3 - B2) + kMargin) - price)   (or tq < 0)
// which implies sqrtf(s) + price > (3 - B2) + kMargin - 7e-7, hence the exact
// value  float(3.0 - sqrtf(s) - price) <= (3 - sqrtf(s) - price) + 1.3e-7
// < B2: the object can change neither best, second best nor (being strictly
// below B2 <= B1) the tie-broken best index.  A cell is skipped with the same
// test on (squared distance to its bounding box, price lower bound); because
// float subtraction/multiply/fma are monotone, every member's own test value
// is >= the cell's, so a skipped cell contains only skippable objects.
constexpr float kMargin = 1e-5f;

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

// Per-person record (32 B): two 16-byte halves.
//   lo = {qx, qy, qz, bits(home chunk)}          the person's point (xyz1); the 16-slot chunk of the sorted objects
//                                                its Hilbert key falls into (emd_index.h: a seed of every search)
//   hi = {bid, prev1, prev2, bits(bidinc)}       slot it last bid on, best /
//        second-best slot of that bid (seed hints), increment of that bid
// Per-object auction state (16 B), next to the object's float4 {x,y,z,price}:
//   {key lo, key hi, owner (-1 = free), -}
//   key = ord(max bid increment this round) << 32 | (winning bidder + 1); 0 = no bid
struct EmdScratch {
  float4 *obj;     // (n) Hilbert-sorted x, y, z, price of xyz2
  int4 *ostate;    // (n) per slot
  float4 *person;  // (2n) per person: lo, hi
  int *perm;       // (n) slot -> original object index
  int *ulist;      // (W x 2n) ping-pong unassigned lists, one pair per workgroup
  u64 *chg;        // (W x kChgCap) {leaf, bits(price bound)} broadcast per round
  int *reserved;   // (kIndexReserve + 4 ints: unused since the leaf index, emd_index.h)
};

// Hand-over record of a cloud (first kernel -> lean kernel), after the barrier granules.
// next_it == 0: nothing to resume.  The members' lists of unassigned persons are at the start
// of their list areas (EmdScratch::ulist).
struct EmdHandover {
  int next_it;     // first round the lean kernel runs
  int utot;        // persons still unassigned
  int unused_[5];  // (the grid geometry of rounds 1-5; the leaf index is a function of n alone: emd_index.h)
  int err;         // the first kernel's internal-error flag
  int cnt[kMaxCluster];   // entries of each member's list
  int nlists;      // members that left a list (the cluster width of the kernel that wrote the record)
  int epoch;       // barrier count of the launches so far (launches that share a set of granules go on counting)
  int first_it;    // round of the FIRST hand-over (kept for the statistics; next_it returns to 0 when the cloud is done)
  int last_width;  // cluster width of the launch that finished the cloud + 16 * that launch's granule set (1, 2)
  int epoch_g;     // gathered-bid rounds run so far (their granules' tags go on counting across launches)
  int pad_[3];
};
static_assert(sizeof(EmdHandover) % 16 == 0, "the scratch tail stays 16-byte granular");

__host__ __device__ inline size_t emd_scratch_per_cloud(int n) {
  // obj 16 + ostate 16 + person 32 + perm 4 + lists 8 x kMaxCluster, + chg + reserve
  return (size_t)n * (68 + 8 * kMaxCluster) + (size_t)kMaxCluster * kChgCap * 8 + (size_t)(kIndexReserve + 4) * 4;
}
// After the per-cloud areas ("tail" of the scratch buffer, zeroed by the host
// before the launch): 256 B of barrier granules per cloud for the first
// kernel, 256 B per cloud for each of the lean kernel's two launches, the hand-over records, then
// the per-cloud statistics {rounds, bids} (last: read by bench.py).
constexpr int kEmdGranuleSets = 3;
constexpr size_t kEmdTailPerCloud = 256 * kEmdGranuleSets + kEmdBidBytes + sizeof(EmdHandover) + 16;
__host__ __device__ inline unsigned long long *emd_granules(char *tail, int b, int cloud, int which) {
  return reinterpret_cast<unsigned long long *>(tail + (size_t)which * b * 256 + (size_t)cloud * 256);
}
// (after the granule sets: the bid areas of the gathered-bid rounds, kEmdBidBytes per cloud)
__host__ __device__ inline unsigned long long *emd_bid_area(char *tail, int b, int cloud) {
  return reinterpret_cast<unsigned long long *>(tail + (size_t)b * 256 * kEmdGranuleSets + (size_t)cloud * kEmdBidBytes);
}
__host__ __device__ inline EmdHandover *emd_handover(char *tail, int b, int cloud) {
  return reinterpret_cast<EmdHandover *>(tail + (size_t)b * (256 * kEmdGranuleSets + kEmdBidBytes)) + cloud;
}
__host__ __device__ inline long long *emd_stats(char *tail, int b, int cloud) {
  return reinterpret_cast<long long *>(tail + (size_t)b * (256 * kEmdGranuleSets + kEmdBidBytes + sizeof(EmdHandover))) + 2 * (size_t)cloud;
}

__device__ __forceinline__ EmdScratch emd_carve(char *base, int n) {
  EmdScratch s;
  s.obj = reinterpret_cast<float4 *>(base);
  s.ostate = reinterpret_cast<int4 *>(base + (size_t)n * 16);
  s.person = reinterpret_cast<float4 *>(base + (size_t)n * 32);
  s.perm = reinterpret_cast<int *>(base + (size_t)n * 64);
  s.ulist = reinterpret_cast<int *>(base + (size_t)n * 68);
  s.chg = reinterpret_cast<u64 *>(base + (size_t)n * (68 + 8 * kMaxCluster));
  s.reserved = reinterpret_cast<int *>(base + (size_t)n * (68 + 8 * kMaxCluster) + (size_t)kMaxCluster * kChgCap * 8);
  return s;
}

// Order-preserving map float -> unsigned (and back); never 0 for a non-NaN.
__device__ __forceinline__ unsigned emd_f2ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float emd_ord2f(unsigned o) {
  return __uint_as_float((o & 0x80000000u) ? (o ^ 0x80000000u) : ~o);
}
// A bidder with increment bi competes for an object whose maximal increment
// is mi (emd_cuda.cu:188).
__device__ __forceinline__ bool emd_in_band(float bi, float mi) {
  return (double)bi - 1e-6 <= (double)mi && (double)mi <= (double)bi + 1e-6;
}
// `old` = the object's key before this bidder's atomic max.  True if a
// DIFFERENT increment close enough to compete was already bid (the test is
// wider than the band: false alarms only cost the explicit GetMax pass).
__device__ __forceinline__ bool emd_band_alarm(u64 old, float inc) {
  if (old == 0ull) return false;
  const float mo = emd_ord2f((unsigned)(old >> 32));
  const double d = (double)mo - (double)inc;
  return mo != inc && d <= 2e-6 && d >= -2e-6;
}

// Reference merge order between two candidates (ORIGINAL object indices) of
// equal value: lexicographically smaller (thread_in_unass, tile, k) wins.
__device__ __attribute__((noinline)) bool emd_precedes(int ka, int kb, int n, int tpu) {
  // tpu <= 0: the caller passes -(unassigned persons of the round) and leaves the two integer
  // divisions behind thread_per_unass (emd_cuda.cu:107-109) to this rare path (lean kernel: they were
  // ~56 instructions at the top of every round of every wave)
  if (tpu <= 0) {
    const int block_cnt = n >> 10;
    const int upb = (-tpu + block_cnt - 1) / block_cnt;
    tpu = 1024 / upb;
  }
  const int tile_a = ka >> 11, tile_b = kb >> 11;
  const int kka = ka & 2047, kkb = kb & 2047;
  const int end_a = min(n - (tile_a << 11), 2048);
  const int end_b = min(n - (tile_b << 11), 2048);
  const int ta = kka / ((end_a + tpu - 1) / tpu);
  const int tb = kkb / ((end_b + tpu - 1) / tpu);
  if (ta != tb) return ta < tb;
  if (tile_a != tile_b) return tile_a < tile_b;
  return kka < kkb;
}

// Correctly rounded sqrtf -- the value the CPU's sqrtf gives, which the parity rests on.  hipcc expands
// __builtin_sqrtf into {scale tiny inputs by 2^32, v_sqrt_f32 (1 ulp), one correction step with two
// fused residuals, unscale, pass 0 / inf through}: 18 vector instructions + 5 wait states.  Squared
// distances are never inf and practically never in (0, 2^-96), so the correction step alone (the same
// one: identical results for x = 0 and x >= 2^-96) runs here, and a wave in which ANY lane holds such a
// tiny positive input takes the compiler's full expansion instead.
__device__ __forceinline__ float emd_sqrtf(float x) {
  if (__builtin_expect(__any((__float_as_uint(x) - 1u) < 0x0F7FFFFFu), 0)) return __builtin_sqrtf(x);   // 0 < x < 2^-96
  float r = __builtin_amdgcn_sqrtf(x);
  const float rd = __int_as_float(__float_as_int(r) - 1), ru = __int_as_float(__float_as_int(r) + 1);
  const float ed = __builtin_fmaf(-rd, r, x), eu = __builtin_fmaf(-ru, r, x);
  r = ed <= 0.f ? rd : r;   // (x = 0: rd is a NaN pattern, ru the smallest denormal; both comparisons are false)
  r = eu > 0.f ? ru : r;
  return r;
}
__device__ __forceinline__ float emd_value(float s, float p) {
  return (float)(3.0 - (double)emd_sqrtf(s) - (double)p);
}
// the same value from the already rounded distance d = sqrtf(s)
__device__ __forceinline__ float emd_value_d(float d, float p) {
  return (float)(3.0 - (double)d - (double)p);
}

// Wave-uniform running state of one bid.
struct BidState {
  float b1, b2;  // best / second-best value
  int bk, b2k;   // their slots (b2k is only a seed hint)
  float tm;      // filter threshold: <= fl(fl(3 - b2) + kMargin)
  float bp;      // (emd_fold<true> only) price of the best object as the search read it
};

// Fold the candidates flagged in `mask` (exact value v and slot k per lane)
// into the uniform state, lowest lane first.
// PC: the best object's price `pw` (per lane, like v and k) is carried along (st.bp).
template <bool PC = false>
__device__ __forceinline__ void emd_fold(BidState &st, unsigned long long mask,
                                         float v, int k, int n, int tpu,
                                         const int *__restrict__ perm, float pw = 0.f) {
  while (mask) {
    const int l = __builtin_ctzll(mask);
    mask &= mask - 1;
    const float vl =
        __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
    const int kl = __builtin_amdgcn_readlane(k, l);
    if (vl > st.b1) {
      st.b2 = st.b1;
      st.b2k = st.bk;
      st.b1 = vl;
      st.bk = kl;
      if constexpr (PC) st.bp = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pw), l));
    } else if (__builtin_expect(vl == st.b1, 0)) {
      st.b2 = st.b1;
      if (emd_precedes(perm[kl], perm[st.bk], n, tpu)) {
        st.b2k = st.bk;
        st.bk = kl;
        if constexpr (PC) st.bp = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pw), l));
      } else {
        st.b2k = kl;
      }
    } else if (vl > st.b2) {
      st.b2 = vl;
      st.b2k = kl;
    }
  }
  // thresholds only ever tighten (the seed may already be tighter)
  st.tm = __builtin_fminf(st.tm, (3.0f - st.b2) + kMargin);
}

__device__ __forceinline__ void top2_insert(float &a1, float &a2, float v) {
  const float lo = __builtin_fminf(a1, v);
  a1 = __builtin_fmaxf(a1, v);
  a2 = __builtin_fmaxf(a2, lo);
}

// DPP move with an identity fill for lanes the control/row mask does not write.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float identity, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v),
                                                    CTRL, ROW_MASK, 0xF, false));
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void top2_dpp_step(float &a1, float &a2) {
  const float o1 = dpp_f32<CTRL, ROW_MASK>(-1e9f, a1);
  const float o2 = dpp_f32<CTRL, ROW_MASK>(-1e9f, a2);
  const float lo = __builtin_fminf(a1, o1);
  a1 = __builtin_fmaxf(a1, o1);
  a2 = __builtin_fmaxf(lo, __builtin_fmaxf(a2, o2));
}

// Second-largest value (with multiplicity) over the wave's per-lane (a1, a2)
// top-2 pairs, using DPP row operations only; every step merges disjoint lane
// sets, masked-out lanes merge with the identity (-1e9, -1e9).  Valid in lane
// 63, returned wave-uniform.
__device__ __forceinline__ float wave_second_largest(float a1, float a2) {
  top2_dpp_step<0xB1, 0xF>(a1, a2);   // quad_perm [1,0,3,2]
  top2_dpp_step<0x4E, 0xF>(a1, a2);   // quad_perm [2,3,0,1]
  top2_dpp_step<0x141, 0xF>(a1, a2);  // row_half_mirror
  top2_dpp_step<0x140, 0xF>(a1, a2);  // row_mirror
  top2_dpp_step<0x142, 0xA>(a1, a2);  // row_bcast15 -> rows 1, 3
  top2_dpp_step<0x143, 0xC>(a1, a2);  // row_bcast31 -> rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a2), 63));
}

// Largest and second-largest (with multiplicity) of the wave's per-lane values.
__device__ __forceinline__ void wave_top2(float v, float &b1, float &b2) {
  float a1 = v, a2 = -1e9f;
  top2_dpp_step<0xB1, 0xF>(a1, a2);
  top2_dpp_step<0x4E, 0xF>(a1, a2);
  top2_dpp_step<0x141, 0xF>(a1, a2);
  top2_dpp_step<0x140, 0xF>(a1, a2);
  top2_dpp_step<0x142, 0xA>(a1, a2);
  top2_dpp_step<0x143, 0xC>(a1, a2);
  b1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a1), 63));
  b2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a2), 63));
}

// max over the wave (uniform result): six v_max_f32 with a DPP operand -- lanes a step does not write keep their own
// value -- spelled in assembly: the compiler expands each step of the intrinsic form into {mov, nop, mov_dpp,
// canonicalising max, max} (30 instructions per reduction, four dependent ones per step).  The s_nop 1 in front
// of every step is the VALU-write -> DPP-read hazard the assembler does not handle inside an asm block.
__device__ __forceinline__ float emd_wave_max(float v) {
  asm volatile(
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Second-largest value (with multiplicity) of the wave's per-lane values: the maximum, then the maximum of the
// rest with one holder of the maximum set aside.  Two DPP reductions of six instructions each instead of
// wave_second_largest's six identity-filled merge steps (~55 instructions).
__device__ __forceinline__ float emd_wave_second(float v) {
  const float t1 = emd_wave_max(v);
  const int l1 = (int)__builtin_ctzll(__ballot(v == t1));
  return emd_wave_max((int)(threadIdx.x & 63) == l1 ? -1e9f : v);
}

// All-gather of two 32-bit payloads among the W workgroups of a cluster; also
// the cluster's barrier.  Every global store the workgroup issued before the
// call is complete (acknowledged write-through) before its granules are
// published.  Granule = one aligned 8-byte {epoch, payload} written by ONE
// sc1 store and polled with sc1 loads: the data is the flag.  Slots are
// double-buffered by epoch parity (a workgroup can be at most one epoch ahead
// of the slowest reader).  Returns false when the wait was abandoned.
// (WM: the granule layout's width and the bound of the poll; W <= WM members take part --
// W is a compile-time constant everywhere but in the lean kernel's planned launch)
template <int WM, bool DRAIN = true>
__device__ __forceinline__ bool emd_cluster_gather_n(u64 *slots, int W, int wg, unsigned epoch,
                                                     const int *p0, const int *p1,
                                                     unsigned *s_gout, int *s_abort, bool same_xcd = false) {
  // DRAIN = false: nothing stored since the last gather has to be visible to
  // the other workgroups before the NEXT draining gather
  if constexpr (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x < kWave) {
    const int lane = threadIdx.x;
    u64 *base = slots + (size_t)(epoch & 1u) * (2 * WM);
    if (lane < 2) {
      const unsigned pv = (unsigned)(lane == 0 ? *p0 : *p1);
      if (same_xcd)  // the pollers share this XCD's L2: no need to write through
        __hip_atomic_store(base + 2 * wg + lane, ((u64)epoch << 32) | pv, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
      else
        __hip_atomic_store(base + 2 * wg + lane, ((u64)epoch << 32) | pv, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    u64 x = (u64)epoch << 32;
    bool done = false;
    for (unsigned spins = 0; spins < kSpinLimit; ++spins) {
      if (lane < 2 * W)
        x = __hip_atomic_load(base + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      done = __all((unsigned)(x >> 32) == epoch);
      if (done) break;
      __builtin_amdgcn_s_sleep(1);
    }
    if (lane < 2 * W) s_gout[lane] = (unsigned)x;
    if (!done && lane == 0) *s_abort = 1;
  }
  __syncthreads();
  return *s_abort == 0;
}
template <int W, bool DRAIN = true>
__device__ __forceinline__ bool emd_cluster_gather(u64 *slots, int wg, unsigned epoch,
                                                   const int *p0, const int *p1,
                                                   unsigned *s_gout, int *s_abort, bool same_xcd = false) {
  return emd_cluster_gather_n<W, DRAIN>(slots, W, wg, epoch, p0, p1, s_gout, s_abort, same_xcd);
}

template <int CTRL>
__device__ __forceinline__ int quad_i32(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}

// LDS-only phase boundary: this wave's LDS traffic has landed, then s_barrier.
// Global stores are NOT waited for.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One step of the wave-wide merge of per-lane exact top-2 {b1, slot k1, b2, slot k2}: the two
// lane sets are disjoint; equal best values are ordered by the reference's rule on ORIGINAL
// object indices (perm).  Lanes the DPP control / row mask does not write merge with nothing.
template <int CTRL, int RM>
__device__ __forceinline__ void top2k_merge_step(float &b1, int &k1, float &b2, int &k2, int n, int tpu,
                                                 const int *__restrict__ perm) {
  const float ob1 = dpp_f32<CTRL, RM>(-1e9f, b1);
  const float ob2 = dpp_f32<CTRL, RM>(-1e9f, b2);
  const int ok1 = __builtin_amdgcn_update_dpp(-1, k1, CTRL, RM, 0xF, false);
  const int ok2 = __builtin_amdgcn_update_dpp(-1, k2, CTRL, RM, 0xF, false);
  const bool tie = ob1 == b1 && ok1 >= 0 && k1 >= 0;
  bool other_first = false;
  if (__any(tie)) {
    if (tie) other_first = emd_precedes(perm[ok1], perm[k1], n, tpu);
  }
  if (ob1 > b1 || other_first) {
    const bool fb = b1 >= ob2;
    b2 = fb ? b1 : ob2;
    k2 = fb ? k1 : ok2;
    b1 = ob1;
    k1 = ok1;
  } else {
    const bool fo = ob1 >= b2;
    k2 = fo ? ok1 : k2;
    b2 = fo ? ob1 : b2;
  }
}

}  // namespace mvp
