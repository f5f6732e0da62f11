// The launch that finishes clouds of at most 4096 points LDS-resident (see emd_resident.h).
#include "emd_resident.h"

namespace mvp {

template <int NMAX>
__global__ __launch_bounds__(kEmdThreads) void emd_resident_kernel(
    int b, int n, float *__restrict__ dist, int *assignment, float eps, int iters, char *scratch) {
  __shared__ ResShared<NMAX> sh;
  emd_resident_body<NMAX>(sh, (int)blockIdx.x, b, n, dist, assignment, eps, iters, scratch);
}

// Finishes the clouds emd_lean.hip's launch stopped for it (hand-over records with next_it != 0).
hipError_t emd_resident_launch(int b, int n, float *dist, int *assignment, float eps, int iters, char *scratch,
                               hipStream_t stream) {
  if (n <= 2048)
    hipLaunchKernelGGL(emd_resident_kernel<2048>, dim3(b), dim3(kEmdThreads), 0, stream, b, n, dist, assignment, eps,
                       iters, scratch);
  else if (n <= kResMaxN)
    hipLaunchKernelGGL(emd_resident_kernel<4096>, dim3(b), dim3(kEmdThreads), 0, stream, b, n, dist, assignment, eps,
                       iters, scratch);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace mvp
