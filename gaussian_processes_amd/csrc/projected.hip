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

// ---- fused E-step in the projected basis (gpfit_estep_projected, api_solve.hip)
// One wave per training point i: lam = a_i . m_b, s_i = A sqrt(f_i), u_i = A^2 f_i lam + A (r_i - f_i)
// (the right-hand side G m + g of utils.py:1431 before its projection a^T); zero on the padding rows.
__global__ __launch_bounds__(256) void estep_proj_rows_kernel(const double* __restrict__ a, int64_t lda, int nb,
                                                               const double* __restrict__ mb, const double* __restrict__ f,
                                                               const double* __restrict__ r, int n, int nrows, double A,
                                                               double* __restrict__ sv, double* __restrict__ u) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= nrows) return;
  double s0 = 0.0;
  if (i < n) {
    const double* row = a + (int64_t)i * lda;
    for (int j = lane; j < nb; j += 64) s0 += row[j] * mb[j];
  }
  for (int o = 32; o > 0; o >>= 1) s0 += __shfl_down(s0, o);
  if (lane == 0) {
    if (i < n) {
      const double fi = f[i];
      sv[i] = A * sqrt(fi);
      u[i] = A * A * fi * s0 + A * (r[i] - fi);
    } else {
      sv[i] = 0.0;
      u[i] = 0.0;
    }
  }
}

// Y[i][j] = s_i aL[i][j], zero padded to [nrows][ld] (and, if asked for, the zero-padded copy aLp of aL itself);
// part[slice][j] = sum over the 32 rows of the slice of aL[i][j] u_i (added up slice by slice afterwards: deterministic)
__global__ __launch_bounds__(256) void estep_proj_scale_kernel(const double* __restrict__ aL, int64_t ldal, int nb, int n,
                                                                const double* __restrict__ sv, const double* __restrict__ u,
                                                                double* __restrict__ Y, double* __restrict__ aLp,
                                                                int64_t ld, int npc, double* __restrict__ part) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= npc) return;
  const int i0 = blockIdx.y * 32;
  double acc = 0.0;
#pragma unroll 8
  for (int q = 0; q < 32; ++q) {
    const int i = i0 + q;
    const double v = (i < n && j < nb) ? aL[(int64_t)i * ldal + j] : 0.0;
    Y[(int64_t)i * ld + j] = sv[i] * v;
    if (aLp) aLp[(int64_t)i * ld + j] = v;
    acc += v * u[i];
  }
  part[(int64_t)blockIdx.y * npc + j] = acc;
}

int launch_estep_proj_rows(const double* a, int64_t lda, int nb, const double* mb, const double* f, const double* r,
                           int n, int nrows, double A, double* sv, double* u, hipStream_t s) {
  hipLaunchKernelGGL(estep_proj_rows_kernel, dim3((nrows + 3) / 4), dim3(256), 0, s, a, lda, nb, mb, f, r, n, nrows, A,
                     sv, u);
  GP_HIP(hipGetLastError());
  return 0;
}
int launch_estep_proj_scale(const double* aL, int64_t ldal, int nb, int n, int nrows, const double* sv, const double* u,
                            double* Y, double* aLp, int64_t ld, int npc, double* part, hipStream_t s) {
  hipLaunchKernelGGL(estep_proj_scale_kernel, dim3((npc + 255) / 256, nrows / 32), dim3(256), 0, s, aL, ldal, nb, n, sv,
                     u, Y, aLp, ld, npc, part);
  GP_HIP(hipGetLastError());
  return 0;
}

// The moments of lambda behind the update, from Z = aL L_W^-T (one wave per training point):
//   lam_m_i = a_i . m_new = Z_i . z1 (z1 = L_W^-1 aL^T u),  lam_var_i = kv0_i + a_i V_new a_i^T = kv0_i + |Z_i|^2
__global__ __launch_bounds__(256) void estep_proj_moments_kernel(const double* __restrict__ Z, int64_t ld, int nb,
                                                                  const double* __restrict__ z1,
                                                                  const double* __restrict__ kv0, int n,
                                                                  double* __restrict__ lam_m, double* __restrict__ lam_var) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const double* row = Z + (int64_t)i * ld;
  double s0 = 0.0, s1 = 0.0;
  for (int j = lane; j < nb; j += 64) {
    const double v = row[j];
    s0 += v * z1[j];
    s1 += v * v;
  }
  for (int o = 32; o > 0; o >>= 1) {
    s0 += __shfl_down(s0, o);
    s1 += __shfl_down(s1, o);
  }
  if (lane == 0) {
    lam_m[i] = s0;
    lam_var[i] = kv0[i] + s1;
  }
}
int launch_estep_proj_moments(const double* Z, int64_t ld, int nb, const double* z1, const double* kv0, int n,
                              double* lam_m, double* lam_var, hipStream_t s) {
  hipLaunchKernelGGL(estep_proj_moments_kernel, dim3((n + 3) / 4), dim3(256), 0, s, Z, ld, nb, z1, kv0, n, lam_m, lam_var);
  GP_HIP(hipGetLastError());
  return 0;
}

}  // namespace gpfit
