// How fast ONE wave can issue fp64 MFMAs on gfx950 (dev tool; calibrates the leaf's cycle budget):
//   hipcc --offload-arch=gfx950 -O3 scripts/scratch/dev_mfma_issue.hip -o gpurun_tmp/mfma_issue && gpurun_tmp/mfma_issue
// chains = 1: every MFMA depends on the one before (same accumulator); chains = 3 / 9: round robin over independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ void k(double* out, long long* cyc, int iters) {
  d4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int CH>
void run(int waves, const char* what) {
  double* out; long long* cyc;
  hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 64 * 8);
  const int iters = 200;
  k<CH><<<1, 64 * waves>>>(out, cyc, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  k<CH><<<1, 64 * waves>>>(out, cyc, iters);
  hipEventRecord(e1, 0); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[64]; hipMemcpy(h, cyc, 64 * 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 4 * CH;
  printf("%-28s waves %d: %.1f counter ticks per MFMA (wave 0), %.1f ns per MFMA by events\n", what, waves, h[0] / n, ms * 1e6 / n);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<1>(1, "dependent chain");
  run<3>(1, "3 chains interleaved");
  run<9>(1, "9 chains interleaved");
  run<3>(4, "3 chains, 4 waves (4 SIMDs)");
  run<3>(5, "3 chains, 5 waves");
  run<3>(8, "3 chains, 8 waves");
  return 0;
}
