// Active-learning utility U = H(r|x,D) - <H(r|f,x)> for a batch of candidate stimuli
// (reference: nd_utility and its helpers, utils.py:413-525; the Lambert W the reference takes
// from scipy on the host, utils.py:464-466, is evaluated on the device).
// One workgroup per candidate; the threads stride over the response counts r_k; the three sums
// over k are reduced in a fixed order (bit-reproducible).
#include "common.h"
#include "kernels.h"

namespace gpfit {

// principal branch, real z >= 0: logarithmic start + Fritsch's quartic iteration on w = ln(z/w)
__device__ __forceinline__ double lambert_w0(double z) {
  if (!(z > 0.0)) return 0.0;
  double w = (z < 2.0) ? z / (1.0 + z) : log(z) - log(log(z));
#pragma unroll 1
  for (int it = 0; it < 6; ++it) {
    const double zn = log(z / w) - w;
    const double q = 2.0 * (1.0 + w) * (1.0 + w + (2.0 / 3.0) * zn);
    const double eps = zn / (1.0 + w) * (q - zn) / (q - 2.0 * zn);
    w = w * (1.0 + eps);
    if (fabs(eps) < 1e-17) break;
  }
  return w;
}

constexpr int UT_THREADS = 128;

__global__ __launch_bounds__(UT_THREADS) void nd_utility_kernel(const double* __restrict__ sigma2,
                                                                const double* __restrict__ mu,
                                                                const double* __restrict__ r, int nr,
                                                                double* __restrict__ U) {
  __shared__ double red[2][UT_THREADS];
  const int i = blockIdx.x;
  const double s2 = sigma2[i], m = mu[i];
  double s_plogp = 0.0, s_plrf = 0.0;
  for (int k = threadIdx.x; k < nr; k += UT_THREADS) {
    double rk = r[k];
    double rs = rk * s2;
    double z = exp(rs + m) * s2;                       // utils.py:447
    double lrf = lgamma(rk + 1.0);                     // utils.py:481
    if (z == INFINITY) {                               // utils.py:450-453, 489-490: overflowing terms
      z = 0.0; rs = 0.0; rk = 0.0; lrf = 0.0;
    }
    const double lam = rs + m - lambert_w0(z);         // utils.py:466
    const double e = exp(lam);
    const double dl = lam - m;
    const double logp = lam * rk - e - dl * dl / (2.0 * s2) - 0.5 * log(e * s2 + 1.0) - lrf;  // utils.py:494
    const double p = exp(logp);
    s_plogp += p * logp;
    s_plrf += p * lrf;
  }
  red[0][threadIdx.x] = s_plogp;
  red[1][threadIdx.x] = s_plrf;
  __syncthreads();
  for (int s = UT_THREADS / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double H_r = -red[0][0];                                               // utils.py:516
    const double H_mean = -exp(m + 0.5 * s2) * (m + s2 - 1.0) + red[1][0];       // utils.py:428
    U[i] = H_r - H_mean;                                                         // utils.py:519
  }
}

int launch_nd_utility(const double* sigma2, const double* mu, int64_t nstar, const double* r, int nr, double* U,
                      hipStream_t s) {
  if (nstar <= 0) return 0;
  hipLaunchKernelGGL(nd_utility_kernel, dim3((unsigned)nstar), dim3(UT_THREADS), 0, s, sigma2, mu, r, nr, U);
  GP_HIP(hipGetLastError());
  return 0;
}

}  // namespace gpfit

using namespace gpfit;

extern "C" int gpfit_nd_utility(void* stream, const double* sigma2, const double* mu, int64_t nstar, const double* r,
                                int nr, double* U) {
  if (!sigma2 || !mu || !r || !U || nstar < 0 || nr <= 0 || nstar > 0x7fffffff) {
    set_error("gpfit_nd_utility: bad argument");
    return -3;
  }
  return launch_nd_utility(sigma2, mu, nstar, r, nr, U, (hipStream_t)stream);
}
