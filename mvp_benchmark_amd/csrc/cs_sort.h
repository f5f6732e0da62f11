// Morton-sorted point sets with bounding boxes per 16 ("tile") and per 1024 ("batch") sorted points: the
// index behind the pruned exact searches (chamfer.hip: nearest neighbour; pn2_query.hip: k nearest).
// chamfer_sort_kernel (chamfer.hip) writes it into caller-provided scratch; see there for the exactness
// argument of the box tests.
#pragma once
#include "common.h"

namespace mvp {

constexpr int kCsThreads = 1024;
constexpr int kCsCells = 4096;  // 16^3
constexpr int kCsTile = 16;
constexpr int kCsBatch = 1024;
constexpr int kCsPad = 0x7fffffff;  // original index of a padding entry

__host__ __device__ inline long long cs_round_up(long long c) { return (c + kCsBatch - 1) / kCsBatch * kCsBatch; }
// bytes of one sorted side holding c points: points + tile boxes + batch boxes
__host__ __device__ inline long long cs_side_bytes(long long c) {
  const long long cp = cs_round_up(c);
  return cp * 16 + cp / kCsTile * 32 + cp / kCsBatch * 32;
}

struct CsSide {
  float4 *pts;   // cp sorted points (padding: +inf coordinates, index kCsPad)
  float4 *tbox;  // 2 per tile: lo, hi
  float4 *bbox;  // 2 per batch: lo, hi
};
__host__ __device__ inline CsSide cs_carve(char *base, long long c) {
  const long long cp = cs_round_up(c);
  CsSide s;
  s.pts = reinterpret_cast<float4 *>(base);
  s.tbox = reinterpret_cast<float4 *>(base + cp * 16);
  s.bbox = reinterpret_cast<float4 *>(base + cp * 16 + cp / kCsTile * 32);
  return s;
}

// squared distance between two axis-aligned boxes (0 if they overlap), with
// the rounding behaviour described above
__device__ __forceinline__ float cs_box_dist(const float (&qlo)[3], const float (&qhi)[3],
                                             const float4 &tlo, const float4 &thi) {
  const float gx = __builtin_fmaxf(__builtin_fmaxf(tlo.x - qhi[0], qlo[0] - thi.x), 0.f);
  const float gy = __builtin_fmaxf(__builtin_fmaxf(tlo.y - qhi[1], qlo[1] - thi.y), 0.f);
  const float gz = __builtin_fmaxf(__builtin_fmaxf(tlo.z - qhi[2], qlo[2] - thi.z), 0.f);
  return sqdist3(gx, gy, gz);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float cs_dpp_max(float v) {
  const float o = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
  return __builtin_fmaxf(v, o);
}
// wave64 maximum with DPP row operations only; result taken from lane 63
__device__ __forceinline__ float cs_wave_max(float v) {
  v = cs_dpp_max<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v = cs_dpp_max<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v = cs_dpp_max<0x141, 0xF>(v);  // row_half_mirror
  v = cs_dpp_max<0x140, 0xF>(v);  // row_mirror
  v = cs_dpp_max<0x142, 0xA>(v);  // row_bcast15 -> rows 1, 3
  v = cs_dpp_max<0x143, 0xC>(v);  // row_bcast31 -> rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Sorts both sides of every cloud pair: xyz1 (b, n1, 3) -> side 0, xyz2 (b, n2, 3) -> side 1 of the cloud's
// scratch area (cs_side_bytes(n1) + cs_side_bytes(n2) bytes per cloud).  chamfer.hip.
void cs_sort_launch(int b, int n1, int n2, const float *xyz1, const float *xyz2, char *scratch, hipStream_t stream);

}  // namespace mvp
