// Element-wise pieces of the fused truncated-rank (B-projected) M-step closure
// (gpfit_fit_eval_projected, fit.hip): moments / rate / likelihood over the N x n projected
// matrices, the adjoint assembly, and the small n x n combinations.  fp64 only (the reference's
// precision; utils.py:31-33).
#include "kernels.h"

namespace gpfit {

// One wave per training point i (utils.py:1090, 1101, 1138 with a = B):
//   lam_m = B_i . m_b,  lam_var = Kvec_i - B_i . Kb_i + aV_i . B_i,  f = exp(A lam_m + A^2/2 lam_var + lambda0)
//   g_m = A (r - f),  g_v = -A^2 f / 2;   block partial sums of r lam_m, r, f -> part[3][gridDim.x]
__global__ __launch_bounds__(256) void proj_moments_kernel(const double* __restrict__ Bp, const double* __restrict__ Kb,
                                                            const double* __restrict__ aV, int64_t ld, int nb,
                                                            const double* __restrict__ mb, const double* __restrict__ Kvec,
                                                            const double* __restrict__ r, int n, double A, double lambda0,
                                                            double* __restrict__ lam_m, double* __restrict__ lam_var,
                                                            double* __restrict__ f, double* __restrict__ gm,
                                                            double* __restrict__ gv, double* __restrict__ part) {
  __shared__ double red[3][4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + w;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  if (i < n) {
    const double* b = Bp + (int64_t)i * ld;
    const double* k = Kb + (int64_t)i * ld;
    const double* a = aV + (int64_t)i * ld;
    for (int j = lane; j < nb; j += 64) {
      const double bj = b[j];
      s0 += bj * mb[j];
      s1 += bj * k[j];
      s2 += a[j] * bj;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    s0 += __shfl_down(s0, o);
    s1 += __shfl_down(s1, o);
    s2 += __shfl_down(s2, o);
  }
  if (lane == 0) {
    double c0 = 0.0, c1 = 0.0, c2 = 0.0;
    if (i < n) {
      const double lm = s0, lv = Kvec[i] - s1 + s2;
      const double fi = exp(A * lm + 0.5 * A * A * lv + lambda0);
      lam_m[i] = lm; lam_var[i] = lv; f[i] = fi;
      gm[i] = A * (r[i] - fi);
      gv[i] = -0.5 * A * A * fi;
      c0 = r[i] * lm; c1 = r[i]; c2 = fi;
    }
    red[0][w] = c0; red[1][w] = c1; red[2][w] = c2;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int q = threadIdx.x;
    part[(int64_t)q * gridDim.x + blockIdx.x] = red[q][0] + red[q][1] + red[q][2] + red[q][3];
  }
}

// out[q] = sum of part[q][0..nblk) in index order (deterministic), q = 0..2
__global__ __launch_bounds__(256) void proj_sum3_kernel(const double* __restrict__ part, int nblk, double* __restrict__ out) {
  __shared__ double red[256];
  for (int q = 0; q < 3; ++q) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += part[(int64_t)q * nblk + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[q] = red[0];
    __syncthreads();
  }
}

// G_a = g_m m_b^T - diag(g_v) K_b + 2 diag(g_v) aV   (N x nb, rows >= n left zero)
__global__ __launch_bounds__(256) void proj_ga_kernel(const double* __restrict__ Kb, const double* __restrict__ aV,
                                                       int64_t ld, int nb, int n, const double* __restrict__ gm,
                                                       const double* __restrict__ gv, const double* __restrict__ mb,
                                                       double* __restrict__ Ga) {
  const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (j >= nb) return;
  const int64_t o = (int64_t)i * ld + j;
  Ga[o] = (i < n) ? gm[i] * mb[j] - gv[i] * Kb[o] + 2.0 * gv[i] * aV[o] : 0.0;
}

// G_Kb = diag(g_v) B - G_a K~_b^-1, in place on the product (rows >= n left zero)
__global__ __launch_bounds__(256) void proj_gkb_kernel(const double* __restrict__ Bp, int64_t ld, int nb, int n,
                                                        const double* __restrict__ gv, double* __restrict__ GaKi) {
  const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (j >= nb) return;
  const int64_t o = (int64_t)i * ld + j;
  GaKi[o] = (i < n) ? gv[i] * Bp[o] - GaKi[o] : 0.0;
}

// G_K~b = 1/2 K~_b^-1 - 1/2 b b^T - 1/2 (K~_b^-1 V_b K~_b^-1) + B^T G_a K~_b^-1     (nb x nb)
__global__ __launch_bounds__(256) void proj_gktb_kernel(const double* __restrict__ Ki, const double* __restrict__ P1,
                                                         const double* __restrict__ P2, int64_t ld, int nb,
                                                         const double* __restrict__ b, double* __restrict__ G) {
  const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (j >= nb) return;
  const int64_t o = (int64_t)i * ld + j;
  G[o] = 0.5 * Ki[o] - 0.5 * b[i] * b[j] - 0.5 * P1[o] + P2[o];
}

// out[0] = sum_i A[i][i], i < n
__global__ __launch_bounds__(256) void proj_trace_kernel(const double* __restrict__ A, int64_t lda, int n, double* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += A[(int64_t)i * lda + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

int launch_proj_moments(const double* Bp, const double* Kb, const double* aV, int64_t ld, int nb, const double* mb,
                        const double* Kvec, const double* r, int n, double A, double lambda0, double* lam_m,
                        double* lam_var, double* f, double* gm, double* gv, double* part, double* out3, hipStream_t s) {
  const int nblk = (n + 3) / 4;
  hipLaunchKernelGGL(proj_moments_kernel, dim3(nblk), dim3(256), 0, s, Bp, Kb, aV, ld, nb, mb, Kvec, r, n, A, lambda0,
                     lam_m, lam_var, f, gm, gv, part);
  hipLaunchKernelGGL(proj_sum3_kernel, dim3(1), dim3(256), 0, s, part, nblk, out3);
  GP_HIP(hipGetLastError());
  return 0;
}
int launch_proj_ga(const double* Kb, const double* aV, int64_t ld, int nb, int n, int np, const double* gm,
                   const double* gv, const double* mb, double* Ga, hipStream_t s) {
  hipLaunchKernelGGL(proj_ga_kernel, dim3((nb + 255) / 256, np), dim3(256), 0, s, Kb, aV, ld, nb, n, gm, gv, mb, Ga);
  GP_HIP(hipGetLastError());
  return 0;
}
int launch_proj_gkb(const double* Bp, int64_t ld, int nb, int n, int np, const double* gv, double* GaKi, hipStream_t s) {
  hipLaunchKernelGGL(proj_gkb_kernel, dim3((nb + 255) / 256, np), dim3(256), 0, s, Bp, ld, nb, n, gv, GaKi);
  GP_HIP(hipGetLastError());
  return 0;
}
int launch_proj_gktb(const double* Ki, const double* P1, const double* P2, int64_t ld, int nb, const double* b, double* G,
                     hipStream_t s) {
  hipLaunchKernelGGL(proj_gktb_kernel, dim3((nb + 255) / 256, nb), dim3(256), 0, s, Ki, P1, P2, ld, nb, b, G);
  GP_HIP(hipGetLastError());
  return 0;
}
int launch_proj_trace(const double* A, int64_t lda, int n, double* out, hipStream_t s) {
  hipLaunchKernelGGL(proj_trace_kernel, dim3(1), dim3(256), 0, s, A, lda, n, out);
  GP_HIP(hipGetLastError());
  return 0;
}

}  // namespace gpfit
