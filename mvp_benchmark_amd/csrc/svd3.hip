// Batched 3x3 SVD + Kabsch rotation for the DCP registration head.
//
// Replaces the per-sample Python loop of SVDHead.forward
// (registration/models/dcp.py:360-373, registration/model_utils.py:229-240):
//   for i in range(B): u, s, v = torch.svd(H[i]); r = v @ u.T
//                      if det(r) < 0: v = v @ diag(1, 1, -1); r = v @ u.T
// i.e. B (B = 128 at BASELINE cfg 5) LAPACK-style calls, 2-4 launches and one
// host synchronisation (`if r_det < 0`) each.  Here: one launch, one lane per
// matrix, no synchronisation.
//
// Method: one-sided Jacobi (Hestenes) on the columns of H in float64 -- plane
// rotations from the right until the columns are mutually orthogonal; then
// sigma_c = |a_c|, u_c = a_c / sigma_c, V = the accumulated rotations; columns
// sorted by descending sigma (torch.svd's order).  Converges quadratically: a
// 3x3 needs <= 5 sweeps; relative accuracy ~1e-15, so the float32 outputs are
// the correctly rounded factors for any H a float32 pipeline produces.
// Rank-deficient H: missing left vectors are completed to an orthonormal basis
// (LAPACK's choice is arbitrary too); the rotation after the reflection fix does
// not depend on that choice when rank(H) >= 2.
#include "common.h"

namespace mvp {

__device__ __forceinline__ void cross3(const double a[3], const double b[3], double c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

__global__ __launch_bounds__(64) void kabsch_svd3_kernel(int b, const float *__restrict__ H, float *__restrict__ R,
                                                         float *__restrict__ Uo, float *__restrict__ So,
                                                         float *__restrict__ Vo, int *__restrict__ flipped) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b) return;
  double A[3][3], V[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      A[r][c] = (double)H[(size_t)i * 9 + r * 3 + c];
      V[r][c] = r == c ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 30; ++sweep) {
    bool rotated = false;
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
      double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        alpha += A[r][p] * A[r][p];
        beta += A[r][q] * A[r][q];
        gamma += A[r][p] * A[r][q];
      }
      if (gamma != 0.0 && fabs(gamma) > 1e-16 * sqrt(alpha * beta)) {
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double tt = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const double ap = A[r][p], aq = A[r][q];
          A[r][p] = cs * ap - sn * aq;
          A[r][q] = sn * ap + cs * aq;
          const double vp = V[r][p], vq = V[r][q];
          V[r][p] = cs * vp - sn * vq;
          V[r][q] = sn * vp + cs * vq;
        }
        rotated = true;
      }
    }
    if (!rotated) break;
  }
  double sg[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) sg[c] = sqrt(A[0][c] * A[0][c] + A[1][c] * A[1][c] + A[2][c] * A[2][c]);
  // sort columns by descending sigma (3-element network)
  auto swapc = [&](int x, int y) {
    if (sg[x] < sg[y]) {
      const double ts = sg[x]; sg[x] = sg[y]; sg[y] = ts;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double ta = A[r][x]; A[r][x] = A[r][y]; A[r][y] = ta;
        const double tv = V[r][x]; V[r][x] = V[r][y]; V[r][y] = tv;
      }
    }
  };
  swapc(0, 1);
  swapc(1, 2);
  swapc(0, 1);
  // left vectors
  double U[3][3];
  const double tol = sg[0] * 1e-14;
  int rank = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const bool ok = sg[c] > tol && sg[c] > 0.0;
    rank += ok ? 1 : 0;
    const double inv = ok ? 1.0 / sg[c] : 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) U[r][c] = A[r][c] * inv;
  }
  if (rank == 0) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) U[r][c] = r == c ? 1.0 : 0.0;
  } else {
    double u0[3] = {U[0][0], U[1][0], U[2][0]}, u1[3], u2[3];
    if (rank == 1) {  // any unit vector orthogonal to u0: cross with the axis u0 is least aligned with
      const double ax = fabs(u0[0]), ay = fabs(u0[1]), az = fabs(u0[2]);
      double e[3] = {0.0, 0.0, 0.0};
      e[ax <= ay && ax <= az ? 0 : (ay <= az ? 1 : 2)] = 1.0;
      cross3(u0, e, u1);
      const double nn = 1.0 / sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
      u1[0] *= nn; u1[1] *= nn; u1[2] *= nn;
      U[0][1] = u1[0]; U[1][1] = u1[1]; U[2][1] = u1[2];
    } else {
      u1[0] = U[0][1]; u1[1] = U[1][1]; u1[2] = U[2][1];
    }
    if (rank <= 2) {
      cross3(u0, u1, u2);
      U[0][2] = u2[0]; U[1][2] = u2[1]; U[2][2] = u2[2];
    }
  }
  // r = v u^T; reflection fix (dcp.py:363-368)
  double Rm[3][3];
  auto make_r = [&]() {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) Rm[r][c] = V[r][0] * U[c][0] + V[r][1] * U[c][1] + V[r][2] * U[c][2];
  };
  make_r();
  const double det = Rm[0][0] * (Rm[1][1] * Rm[2][2] - Rm[1][2] * Rm[2][1]) -
                     Rm[0][1] * (Rm[1][0] * Rm[2][2] - Rm[1][2] * Rm[2][0]) +
                     Rm[0][2] * (Rm[1][0] * Rm[2][1] - Rm[1][1] * Rm[2][0]);
  const int flip = det < 0.0 ? 1 : 0;
  // U, S, V are returned as torch.svd would (before the fix); R with it applied
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (Uo) Uo[(size_t)i * 9 + r * 3 + c] = (float)U[r][c];
      if (Vo) Vo[(size_t)i * 9 + r * 3 + c] = (float)V[r][c];
    }
  if (So) {
    So[(size_t)i * 3 + 0] = (float)sg[0];
    So[(size_t)i * 3 + 1] = (float)sg[1];
    So[(size_t)i * 3 + 2] = (float)sg[2];
  }
  if (flip) {
    V[0][2] = -V[0][2];
    V[1][2] = -V[1][2];
    V[2][2] = -V[2][2];
    make_r();
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) R[(size_t)i * 9 + r * 3 + c] = (float)Rm[r][c];
  if (flipped) flipped[i] = flip;
}

}  // namespace mvp

using namespace mvp;

extern "C" int mvp_kabsch_svd3(int b, const float *H, float *R, float *U, float *S, float *V, int *flipped,
                               void *stream) {
  if (b < 0) return MVP_EBADSHAPE;
  if (b == 0) return MVP_OK;
  if (!H || !R) return MVP_EBADARG;
  hipLaunchKernelGGL(kabsch_svd3_kernel, dim3((b + 63) / 64), dim3(64), 0, as_stream(stream), b, H, R, U, S, V,
                     flipped);
  return check_launch("mvp_kabsch_svd3");
}
