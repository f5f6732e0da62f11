// Latency microbenchmarks on MI355X: dependent global loads (L2-resident and
// MALL/HBM-resident), global store + vmcnt drain, global atomic, LDS read,
// s_barrier with 16 waves, ds_bpermute, DPP.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

__global__ void chase(const int* __restrict__ p, int steps, long long* out, int* sink) {
  int i = threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  for (int s = 0; s < steps; ++s) i = __builtin_nontemporal_load(&p[i]) ;
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[0] = (t1 - t0) / steps; sink[0] = i; }
}
__global__ void chase_plain(const int* p, int steps, long long* out, int* sink) {
  int i = threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  for (int s = 0; s < steps; ++s) i = p[i];
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[0] = (t1 - t0) / steps; sink[0] = i; }
}
__global__ void store_drain(int* p, int steps, long long* out) {
  long long t0 = __builtin_readcyclecounter();
  for (int s = 0; s < steps; ++s) {
    p[(threadIdx.x * 997 + s * 64) & 0xfffff] = s;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = (t1 - t0) / steps;
}
__global__ void atomic_lat(int* p, int steps, long long* out, int* sink) {
  int acc = 0;
  long long t0 = __builtin_readcyclecounter();
  for (int s = 0; s < steps; ++s) acc += atomicMax(&p[(threadIdx.x * 64 + acc) & 0xffff], s);
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[0] = (t1 - t0) / steps; sink[0] = acc; }
}
__global__ void atomic_noret(int* p, int steps, long long* out) {
  long long t0 = __builtin_readcyclecounter();
  for (int s = 0; s < steps; ++s) { atomicMax(&p[(threadIdx.x * 64 + s) & 0xffff], s); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = (t1 - t0) / steps;
}
__global__ void barrier_lat(int steps, long long* out) {
  long long t0 = __builtin_readcyclecounter();
  for (int s = 0; s < steps; ++s) __syncthreads();
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = (t1 - t0) / steps;
}
__global__ void lds_lat(int steps, long long* out, int* sink) {
  __shared__ int a[1024];
  a[threadIdx.x] = (threadIdx.x * 7 + 1) & 1023;
  __syncthreads();
  int i = threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  for (int s = 0; s < steps; ++s) i = a[i];
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[0] = (t1 - t0) / steps; sink[0] = i; }
}
__global__ void bperm_lat(int steps, long long* out, int* sink) {
  int v = threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  for (int s = 0; s < steps; ++s) v = __shfl_xor(v, 1 + (s & 31), 64) + 1;
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[0] = (t1 - t0) / steps; sink[0] = v; }
}
__global__ void valu_dep(int steps, long long* out, float* sink) {
  float v = threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  for (int s = 0; s < steps; ++s) v = __builtin_fmaf(v, 1.0001f, 0.5f);
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[0] = (t1 - t0) * 100 / steps; sink[0] = v; }
}

int main() {
  long long* out; int* sink; hipMalloc(&out, 64); hipMalloc(&sink, 64);
  auto get = [&]() { long long h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); return h; };
  for (size_t bytes : {size_t(64) << 10, size_t(2) << 20, size_t(24) << 20, size_t(1) << 30}) {
    size_t n = bytes / 4;
    std::vector<int> perm(n); std::iota(perm.begin(), perm.end(), 0);
    std::mt19937 g(1); std::shuffle(perm.begin(), perm.end(), g);
    std::vector<int> nxt(n); for (size_t i = 0; i < n; ++i) nxt[perm[i]] = perm[(i + 1) % n];
    int* d; hipMalloc(&d, bytes); hipMemcpy(d, nxt.data(), bytes, hipMemcpyHostToDevice);
    for (int waves : {1, 16}) {
      hipLaunchKernelGGL(chase_plain, 1, 64 * waves, 0, 0, d, 2000, out, sink); hipDeviceSynchronize();
      hipLaunchKernelGGL(chase_plain, 1, 64 * waves, 0, 0, d, 2000, out, sink); hipDeviceSynchronize();
      printf("dependent global load, %zu KB footprint, %d waves: %lld cycles/load\n", bytes >> 10, waves, get());
    }
    hipFree(d);
  }
  int* buf; hipMalloc(&buf, 4 << 20); hipMemset(buf, 0, 4 << 20);
  for (int waves : {1, 16}) {
    hipLaunchKernelGGL(store_drain, 1, 64 * waves, 0, 0, buf, 1000, out); hipDeviceSynchronize();
    printf("global store + vmcnt(0) drain, %d waves: %lld cycles\n", waves, get());
    hipLaunchKernelGGL(atomic_lat, 1, 64 * waves, 0, 0, buf, 1000, out, sink); hipDeviceSynchronize();
    printf("global atomicMax with return (dependent), %d waves: %lld cycles\n", waves, get());
    hipLaunchKernelGGL(atomic_noret, 1, 64 * waves, 0, 0, buf, 1000, out); hipDeviceSynchronize();
    printf("global atomicMax no return + drain, %d waves: %lld cycles\n", waves, get());
    hipLaunchKernelGGL(barrier_lat, 1, 64 * waves, 0, 0, 1000, out); hipDeviceSynchronize();
    printf("__syncthreads, %d waves: %lld cycles\n", waves, get());
    hipLaunchKernelGGL(lds_lat, 1, 64 * waves, 0, 0, 2000, out, sink); hipDeviceSynchronize();
    printf("dependent LDS read, %d waves: %lld cycles\n", waves, get());
    hipLaunchKernelGGL(bperm_lat, 1, 64 * waves, 0, 0, 2000, out, sink); hipDeviceSynchronize();
    printf("dependent ds_bpermute(+add), %d waves: %lld cycles\n", waves, get());
    hipLaunchKernelGGL(valu_dep, 1, 64 * waves, 0, 0, 4000, out, (float*)sink); hipDeviceSynchronize();
    printf("dependent v_fma_f32, %d waves: %lld /100 cycles\n", waves, get());
  }
  return 0;
}
