/* Arithmetic primitives of oracle/emulator.py -- TEST INFRASTRUCTURE (only tests/ may use it).
 *
 * The emulator replays the reference's CUDA grids thread by thread in Python; the only thing it
 * does not do in Python is float32 arithmetic with fused multiply-adds (NumPy has no fmaf).  These
 * few loops are that arithmetic and nothing else -- no search, no merge, no schedule -- written
 * independently of mvp_oracle.c.  nvcc's default -fmad=true contracts a*a + b*b + c*c into
 * fma(c, c, fma(b, b, a*a)) (the expression is parsed left to right, the last addition takes the
 * fused multiply); gcc is kept from contracting anything else with -ffp-contract=off. */
#include <math.h>

/* Bid (emd_cuda.cu:141-146): d[k] = 3.0 - sqrtf(x2*x2 + y2*y2 + z2*z2) - price[k] for the
 * objects k = lo .. hi-1 of a shared-memory tile; `3.0` is a double literal, so the subtraction
 * chain runs in double and the result is narrowed on assignment to `float d`. */
void emu_bid_values(int lo, int hi, float x1, float y1, float z1, const float *xyz2_buf,
                    const float *price_buf, float *d) {
  for (int k = lo; k < hi; ++k) {
    const float x2 = xyz2_buf[k * 3 + 0] - x1;
    const float y2 = xyz2_buf[k * 3 + 1] - y1;
    const float z2 = xyz2_buf[k * 3 + 2] - z1;
    const float s = fmaf(z2, z2, fmaf(y2, y2, x2 * x2));
    d[k - lo] = (float)(3.0 - (double)sqrtf(s) - (double)price_buf[k]);
  }
}

/* furthest_point_sampling_kernel (furthest_point_sample_cuda.cu:64-65): squared distance of the
 * points k = first, first + stride, ... < n to (x1, y1, z1). */
void emu_fps_sqdist(int first, int stride, int n, float x1, float y1, float z1, const float *xyz, float *d) {
  int j = 0;
  for (int k = first; k < n; k += stride, ++j) {
    const float dx = xyz[k * 3 + 0] - x1, dy = xyz[k * 3 + 1] - y1, dz = xyz[k * 3 + 2] - z1;
    d[j] = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
  }
}

/* CalcDist (emd_cuda.cu:217-226) */
float emu_calc_dist(const float *a, const float *b) {
  const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}
