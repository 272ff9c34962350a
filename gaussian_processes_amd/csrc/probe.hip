// On-box microbenchmarks used to confirm the roofline constants (fp64 MFMA issue rate and
// HBM stream bandwidth) that bench.py reports fractions of.
#include "common.h"
#include "gpfit_mi355x.h"

namespace gpfit {
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_f64_probe_kernel(double* out, int iters, double seed) {
  v4d acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = v4d{0.0, 0.0, 0.0, 0.0};
  double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

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
