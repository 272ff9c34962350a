// Cost of a barrier among G persistent workgroups (atomic counter in global memory, agent-scope fences), alone
// and beside a memory-heavy kernel on another stream: the synchronisation primitive a multi-workgroup kernel for
// the bottom levels of the Cholesky recursion would be built on.  Every spin loop is bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ __forceinline__ bool cluster_barrier(unsigned* ctr, unsigned target, int* err) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();                       // release: this workgroup's writes before the arrival
    atomicAdd(ctr, 1u);
    int spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) { *err = 1; ok = false; break; }
    }
    __threadfence();                       // acquire: the others' writes after the arrival
  }
  __syncthreads();
  return ok;
}
// each round: every workgroup writes a 32 x 32 tile of doubles that the NEXT workgroup reads in the next round
__global__ __launch_bounds__(256) void rounds_kernel(unsigned* ctr, int* err, double* buf, int rounds, long long* cycles) {
  const int G = gridDim.x, w = blockIdx.x;
  long long t0 = wall_clock64();
  double acc = 0;
  for (int r = 0; r < rounds; ++r) {
    double* mine = buf + (size_t)((r & 1) * G + w) * 1024;
    for (int e = threadIdx.x; e < 1024; e += 256) mine[e] = r + w + e * 1e-3 + acc * 1e-9;
    if (!cluster_barrier(ctr, (unsigned)(r + 1) * G, err)) return;
    const double* other = buf + (size_t)((r & 1) * G + (w + 1) % G) * 1024;
    for (int e = threadIdx.x; e < 1024; e += 256) acc += other[e];
  }
  if (threadIdx.x == 0) cycles[w] = wall_clock64() - t0;
  if (acc == 12345.678) buf[0] = acc;
  // the last workgroup to leave resets the counter for the next launch
  __syncthreads();
  if (threadIdx.x == 0) { __threadfence(); if (atomicAdd(ctr + 1, 1u) == (unsigned)G - 1) { ctr[0] = 0; ctr[1] = 0; } }
}
__global__ void noise_kernel(double* a, size_t n, int iters) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (int it = 0; it < iters; ++it)
    for (size_t j = i; j < n; j += (size_t)gridDim.x * blockDim.x) a[j] = a[j] * 1.0000001 + 1e-9;
}
int main() {
  unsigned* ctr; int* err; double* buf; long long* cyc; double* big;
  hipMalloc(&ctr, 64); hipMemset(ctr, 0, 64); hipMalloc(&err, 4); hipMemset(err, 0, 4);
  hipMalloc(&buf, 2 * 64 * 1024 * 8); hipMalloc(&cyc, 64 * 8);
  const size_t nbig = (size_t)1 << 28; hipMalloc(&big, nbig * 8); hipMemset(big, 0, nbig * 8);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int rounds = 2000;
  for (int noise = 0; noise < 2; ++noise)
    for (int G : {4, 8, 16, 32, 64}) {
      if (noise) hipLaunchKernelGGL(noise_kernel, dim3(2048), dim3(256), 0, s2, big, nbig, 6);
      hipEventRecord(e0, s1);
      hipLaunchKernelGGL(rounds_kernel, dim3(G), dim3(256), 0, s1, ctr, err, buf, rounds, cyc);
      hipEventRecord(e1, s1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      int herr; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
      printf("%s G = %2d: %.2f us per round (write 8 KiB tile, barrier, read neighbour's tile)%s\n", noise ? "beside a streaming kernel," : "alone,                   ",
             G, ms * 1e3 / rounds, herr ? "  [TIMEOUT]" : "");
      hipDeviceSynchronize();
    }
  return 0;
}
