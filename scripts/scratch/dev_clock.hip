// Sustained fp64 MFMA issue probe with the shader clock read beside it:
//   clock64()      = s_memtime     (shader clock cycles)
//   wall_clock64() = s_memrealtime (constant 100 MHz)
// Prints achieved TFLOP/s and the effective shader clock for runs of growing length, with data-dependent
// operands (random, non-zero) so the matrix pipes toggle as they do in a real GEMM.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4d __attribute__((ext_vector_type(4)));
#define GP_MFMA(acc) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
__global__ __launch_bounds__(256) void probe(double* out, long long* clk, int iters, const double* seed) {
  v4d c[16];
  for (int i = 0; i < 16; ++i) c[i] = v4d{0, 0, 0, 0};
  double a = seed[threadIdx.x], b = seed[256 + threadIdx.x];
  long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) GP_MFMA(c[i]);
  }
  long long t1 = clock64(), w1 = wall_clock64();
  v4d s = c[0];
  for (int i = 1; i < 16; ++i) s += c[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
int main(int argc, char** argv) {
  int blocks = argc > 1 ? atoi(argv[1]) : 1024;
  double *out, *seed; long long* clk;
  hipMalloc(&out, blocks * 256 * 8); hipMalloc(&clk, blocks * 16); hipMalloc(&seed, 512 * 8);
  double h[512]; srand(1); for (int i = 0; i < 512; ++i) h[i] = (rand() / (double)RAND_MAX - 0.5) * 1e-3;
  hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int iters : {200, 2000, 20000, 100000, 100000, 100000}) {
    hipEventRecord(e0); hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, out, clk, iters, seed); hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    double fl = (double)blocks * 4 * iters * 16 * 2048.0;
    printf("blocks %d iters %6d: %8.3f ms  %6.2f TFLOP/s   block 0: %lld shader cycles in %lld x 10 ns -> %.3f GHz; issue fraction %.3f (64 cycles per MFMA per SIMD, %d waves per SIMD)\n",
           blocks, iters, ms, fl / ms / 1e9, hc[0], hc[1], hc[0] / (hc[1] * 10.0),
           (double)iters * 16 * 64 * (blocks / 256) / hc[0], blocks / 256);
  }
  return 0;
}
