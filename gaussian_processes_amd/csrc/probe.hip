// On-box microbenchmarks used to confirm the roofline constants (fp64 MFMA issue rate and
// HBM stream bandwidth) that bench.py reports fractions of.
#include "common.h"
#include "gpfit_mi355x.h"

namespace gpfit {
typedef double v4d __attribute__((ext_vector_type(4)));

// Back-to-back issue of v_mfma_f64_16x16x4_f64 on 16 independent accumulators per wave, written
// in inline asm so the compiler cannot shuffle the accumulators between register files.
#define GP_MFMA(acc) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
__global__ __launch_bounds__(256) void mfma_f64_probe_kernel(double* out, int iters, double seed) {
  v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  v4d c8 = c0, c9 = c0, c10 = c0, c11 = c0, c12 = c0, c13 = c0, c14 = c0, c15 = c0;
  double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
    GP_MFMA(c0); GP_MFMA(c1); GP_MFMA(c2); GP_MFMA(c3); GP_MFMA(c4); GP_MFMA(c5); GP_MFMA(c6); GP_MFMA(c7);
    GP_MFMA(c8); GP_MFMA(c9); GP_MFMA(c10); GP_MFMA(c11); GP_MFMA(c12); GP_MFMA(c13); GP_MFMA(c14); GP_MFMA(c15);
  }
  v4d s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + c8 + c9 + c10 + c11 + c12 + c13 + c14 + c15;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
#undef GP_MFMA

__global__ void stream_copy_kernel(const double2* __restrict__ in, double2* __restrict__ out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i];
}
}  // namespace gpfit

extern "C" {

int gpfit_probe_mfma_f64(void* stream, double* scratch /* >= blocks*256 doubles */, int blocks, int iters) {
  hipLaunchKernelGGL(gpfit::mfma_f64_probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, scratch,
                     iters, 1.0);
  GP_HIP(hipGetLastError());
  return 0;
}

int gpfit_probe_stream_copy(void* stream, const double* in, double* out, int64_t n_doubles) {
  hipLaunchKernelGGL(gpfit::stream_copy_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream,
                     (const double2*)in, (double2*)out, n_doubles / 2);
  GP_HIP(hipGetLastError());
  return 0;
}
}
