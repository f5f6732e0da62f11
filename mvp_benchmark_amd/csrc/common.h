// Shared device/host helpers for libmvpops (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mvpops.h"

namespace mvp {

constexpr int kWave = 64;

// Canonical squared length: fma(c,c, fma(b,b, a*a)).  Same chain as the CPU
// oracle (oracle/mvp_oracle.c sqdist3); built with -ffp-contract=off so the
// compiler neither adds nor removes a contraction.
__device__ __forceinline__ float sqdist3(float ax, float ay, float az) {
  return __builtin_fmaf(az, az, __builtin_fmaf(ay, ay, ax * ax));
}

// Records the last HIP launch error for mvp_last_hip_error().
int check_launch(const char *what);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace mvp
