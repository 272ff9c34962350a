// C-ABI entry points built on the recursive MFMA Cholesky: a standalone factorisation
// (log_det, general solves of the drop-in module), the fused E-step Newton update and the
// one-pass firing-rate-parameter evaluation.
#include "context.h"
#include "gpfit_mi355x.h"

#include <cmath>

using namespace gpfit;

#define GP_TRY(expr)            \
  do {                          \
    int _rc = (expr);           \
    if (_rc != 0) return _rc;   \
  } while (0)

static int gemm_full(hipStream_t s, int ak, int bk, int M, int N, int K, double alpha, const double* A, int64_t lda,
                     const double* B, int64_t ldb, double beta, double* C, int64_t ldc, int lower, int at, int bt,
                     int walk, void* sk_ws) {
  GemmArgs g{};
  g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta; g.a_kmajor = ak; g.b_kmajor = bk;
  g.out_lower = lower; g.a_tri = at; g.b_tri = bt; g.batch = 1; g.split_k = 1; g.reverse = walk;
  g.sk_ws = sk_ws;
  return launch_gemm(g, s);
}

namespace {
// y_i = sum_{j <= i} M[i][j] x_j for a lower-triangular row-major M (one wave per row)
__global__ __launch_bounds__(256) void append_trmv_kernel(const double* __restrict__ M, int64_t ld, int n,
                                                           const double* __restrict__ x, double* __restrict__ y) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  double acc = 0.0;
  for (int j = lane; j <= row; j += 64) acc += M[(int64_t)row * ld + j] * x[j];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if (lane == 0) y[row] = acc;
}
// w_j = sum_{i >= j} M[i][j] l_i (a 64-column strip per workgroup, rows walked in four interleaved
// groups, summed in a fixed order: deterministic); also out[0] = l . l from workgroup 0
__global__ __launch_bounds__(256) void append_trmv_t_kernel(const double* __restrict__ M, int64_t ld, int n,
                                                             const double* __restrict__ l, double* __restrict__ w,
                                                             double* __restrict__ out) {
  __shared__ double part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  double acc = 0.0;
  if (c < n)
    for (int i = c + g; i < n; i += 4) acc += M[(int64_t)i * ld + c] * l[i];
  part[g][threadIdx.x & 63] = acc;
  __syncthreads();
  if (g == 0 && c < n) w[c] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
  if (blockIdx.x == 0) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += l[i] * l[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
  }
}
// row n of both factors: L[n][:] = (l, lambda), Linv[n][:] = (-w / lambda, 1 / lambda); column n zeroed above
__global__ __launch_bounds__(256) void append_write_kernel(double* __restrict__ L, int64_t ldl, double* __restrict__ Li,
                                                            int64_t ldi, int n, const double* __restrict__ l,
                                                            const double* __restrict__ w, const double* __restrict__ kcol,
                                                            double* __restrict__ out, int* __restrict__ info) {
  const double p = kcol[n] - out[0];           // Schur complement of the new diagonal entry
  const bool bad = !(p > 0.0);
  const double lam = bad ? 1.0 : sqrt(p);
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j == 0) {
    out[1] = lam;
    if (bad) atomicCAS(info, 0, n + 1);
  }
  if (j < n) {
    L[(int64_t)n * ldl + j] = l[j];
    Li[(int64_t)n * ldi + j] = -w[j] / lam;
    L[(int64_t)j * ldl + n] = 0.0;
    Li[(int64_t)j * ldi + n] = 0.0;
  } else if (j == n) {
    L[(int64_t)n * ldl + n] = lam;
    Li[(int64_t)n * ldi + n] = 1.0 / lam;
  }
}
}  // namespace

extern "C" {

int gpfit_potrf_append(gpfit_ctx* c, void* stream, double* L, int64_t ldl, double* Linv, int64_t ldi, int64_t n,
                       const double* kcol, double* logdet_inout_host, int* info_host) {
  if (!c || !L || !Linv || !kcol || n <= 0 || ldl <= n || ldi <= n) {
    set_error("gpfit_potrf_append: bad argument (the factors need room for row and column n: ld > n)");
    return -3;
  }
  GP_CTX_ENTER(c, "gpfit_potrf_append");
  hipStream_t s = (hipStream_t)stream;
  if (n + 1 > c->np_cap) {
    set_error("gpfit_potrf_append: matrix larger than the context capacity");
    return -3;
  }
  const int ni = (int)n;
  double* l = c->yv;       // L^-1 k
  double* w = c->bv;       // L^-T l
  GP_HIP(hipMemsetAsync(c->info, 0, 4 * sizeof(int), s));
  hipLaunchKernelGGL(append_trmv_kernel, dim3((ni + 3) / 4), dim3(256), 0, s, Linv, ldi, ni, kcol, l);
  hipLaunchKernelGGL(append_trmv_t_kernel, dim3((ni + 63) / 64), dim3(256), 0, s, Linv, ldi, ni, l, w, c->scal + 48);
  hipLaunchKernelGGL(append_write_kernel, dim3((ni + 256) / 256), dim3(256), 0, s, L, ldl, Linv, ldi, ni, l, w, kcol,
                     c->scal + 48, c->info);
  GP_HIP(hipGetLastError());
  GP_HIP(hipMemcpyAsync(c->scal_host + 48, c->scal + 48, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
  GP_HIP(hipMemcpyAsync(c->info_host, c->info, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  GP_HIP(hipStreamSynchronize(s));
  if (info_host) *info_host = c->info_host[0];
  if (c->info_host[0] != 0) {
    set_error("gpfit_potrf_append: the extended matrix is not positive definite");
    return c->info_host[0];
  }
  if (logdet_inout_host) *logdet_inout_host += 2.0 * std::log(c->scal_host[49]);
  return 0;
}

int gpfit_potrf(gpfit_ctx* c, void* stream, const double* A, int64_t lda, int64_t n, double* L, int64_t ldl,
                double* Linv, int64_t ldi, double* logdet_host, int* info_host) {
  if (!c || !A || n <= 0) {
    set_error("gpfit_potrf: bad argument");
    return -3;
  }
  GP_CTX_ENTER(c, "gpfit_potrf");
  hipStream_t s = (hipStream_t)stream;
  const int np = (int)round_up(n, TILE);
  if (np > c->np_cap) {
    set_error("gpfit_potrf: matrix larger than the context capacity");
    return -3;
  }
  const int64_t ld = np;
  GP_HIP(hipMemsetAsync(c->info, 0, 4 * sizeof(int), s));
  GP_TRY(launch_pack_lower(A, lda, (int)n, c->Kbuf, ld, np, s));
  CholBufs b{c->Kbuf, c->Lbuf, c->Libuf, c->Tmp, ld, c->info, 0, c->sk_ws[0]};
  GP_TRY(potrf_rec(b, 0, np, Linv != nullptr, s));
  GP_TRY(launch_logdet(c->Lbuf, ld, (int)n, c->scal + 3, s));
  if (L) GP_TRY(launch_unpack_tri(c->Lbuf, ld, (int)n, L, ldl, s));
  if (Linv) GP_TRY(launch_unpack_tri(c->Libuf, ld, (int)n, Linv, ldi, s));
  GP_HIP(hipMemcpyAsync(c->scal_host, c->scal, 8 * sizeof(double), hipMemcpyDeviceToHost, s));
  GP_HIP(hipMemcpyAsync(c->info_host, c->info, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  GP_HIP(hipStreamSynchronize(s));
  if (logdet_host) *logdet_host = c->scal_host[3];
  if (info_host) *info_host = c->info_host[0];
  if (c->info_host[0] != 0) {
    set_error("gpfit_potrf: matrix is not positive definite");
    return c->info_host[0];
  }
  return 0;
}

int gpfit_estep(gpfit_ctx* c, void* stream, const double* K, int64_t ldk, int64_t N, const double* r,
                const double* m, const double* f, double logA, double* m_new, double* V_new, int64_t ldv) {
  if (!c || !K || !r || !m || !f || !m_new || !V_new || N <= 0) {
    set_error("gpfit_estep: bad argument");
    return -3;
  }
  GP_CTX_ENTER(c, "gpfit_estep");
  hipStream_t s = (hipStream_t)stream;
  const int n = (int)N, np = (int)round_up(N, TILE);
  if (np > c->np_cap) {
    set_error("gpfit_estep: problem larger than the context capacity");
    return -3;
  }
  const int64_t ld = np;
  const double A = std::exp(logA);
  double* sv = c->yv;
  double* rhs = c->bv;
  GP_HIP(hipMemsetAsync(c->info, 0, 4 * sizeof(int), s));
  GP_TRY(launch_estep_prep(f, r, m, n, np, A, sv, rhs, s));
  // M = I + S K S (lower), SK = S K (dense), Kl = K (lower)
  GP_TRY(launch_estep_build(K, ldk, n, np, sv, c->Kbuf, c->Zbuf, c->Wbuf, ld, s));
  CholBufs b{c->Kbuf, c->Lbuf, c->Libuf, c->Tmp, ld, c->info, 0, c->sk_ws[0]};
  // T = L_M^-1 (S K)          lower x dense                         N^3
  if (np >= 2 * TILE) {
    // block-wise, so that the off-diagonal block of L_M^-1 is never formed (N^3/4 less):
    //   T1 = [L^-1]11 B1 ,  T2 = [L^-1]22 (B2 - L21 T1)        with B = S K
    const int kt = np / TILE;
    const int n1 = ((kt + 1) / 2) * TILE, n2 = np - n1;
    GP_TRY(potrf_rec(b, 0, np, 2, s));
    GP_TRY(gemm_full(s, 0, 1, n1, np, n1, 1.0, c->Libuf, ld, c->Zbuf, ld, 0.0, c->Abuf, ld, 0, 1, 0, 1, c->sk_ws[0]));
    GP_TRY(gemm_full(s, 0, 1, n2, np, n1, -1.0, c->Lbuf + (int64_t)n1 * ld, ld, c->Abuf, ld, 1.0,
                     c->Zbuf + (int64_t)n1 * ld, ld, 0, 0, 0, 0, c->sk_ws[0]));
    GP_TRY(gemm_full(s, 0, 1, n2, np, n2, 1.0, c->Libuf + (int64_t)n1 * ld + n1, ld, c->Zbuf + (int64_t)n1 * ld, ld, 0.0,
                     c->Abuf + (int64_t)n1 * ld, ld, 0, 1, 0, 1, c->sk_ws[0]));
  } else {
    GP_TRY(potrf_rec(b, 0, np, 1, s));
    GP_TRY(gemm_full(s, 0, 1, np, np, np, 1.0, c->Libuf, ld, c->Zbuf, ld, 0.0, c->Abuf, ld, 0, 1, 0, 1, c->sk_ws[0]));
  }
  // V = K - T^T T             lower tiles only                      N^3
  GP_TRY(gemm_full(s, 1, 1, np, np, np, -1.0, c->Abuf, ld, c->Abuf, ld, 1.0, c->Wbuf, ld, 1, 0, 0, 0, c->sk_ws[0]));
  // m_new = V (A^2 f o m + A (r - f))                                utils.py:1431
  GP_TRY(launch_symv_lower(c->Wbuf, ld, n, rhs, c->tvec, s));
  GP_HIP(hipMemcpyAsync(m_new, c->tvec, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
  GP_TRY(launch_unpack_sym(c->Wbuf, ld, n, V_new, ldv, s));  // symmetric by construction (utils.py:1438)
  GP_HIP(hipMemcpyAsync(c->info_host, c->info, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  GP_HIP(hipStreamSynchronize(s));
  if (c->info_host[0] != 0) {
    set_error("gpfit_estep: I + S K S is not positive definite (is K_tilde symmetric positive definite?)");
    return c->info_host[0];
  }
  return 0;
}

int gpfit_estep_projected(gpfit_ctx* c, void* stream, const double* a, int64_t lda, const double* aL, int64_t ldal,
                          const double* L, int64_t ldl, int64_t N, int64_t nb, const double* r, const double* m,
                          const double* f, double logA, double* m_new, double* V_new, int64_t ldv, const double* kv0,
                          double* lam_m_out, double* lam_var_out) {
  const bool want_moments = kv0 && lam_m_out && lam_var_out;
  if (!c || !a || !aL || !L || !r || !m || !f || !m_new || !V_new || N <= 0 || nb <= 0 || lda < nb || ldal < nb ||
      ldl < nb || ldv < nb || (!want_moments && (kv0 || lam_m_out || lam_var_out))) {
    set_error("gpfit_estep_projected: bad argument");
    return -3;
  }
  GP_CTX_ENTER(c, "gpfit_estep_projected");
  hipStream_t s = (hipStream_t)stream;
  const int n = (int)N, k = (int)nb;
  const int nrows = (int)round_up(N, TILE), npc = (int)round_up(nb, TILE);
  if (nrows > c->np_cap || npc > c->np_cap) {
    set_error("gpfit_estep_projected: problem larger than the context capacity");
    return -3;
  }
  const int64_t ld = npc;
  const double A = std::exp(logA);
  double *sv = c->yv, *u = c->bv, *t2 = c->tvec, *z1 = c->mpad, *z = c->rpad, *mo = c->hvec;
  double *Y = c->Tbuf, *Lp = c->Wbuf, *P = c->Abuf, *V = c->Zbuf, *part = c->TmpV, *aLp = c->LiVbuf, *Zm = c->Cos;
  c->lv_valid = false; c->lv32_valid = false;   // the work matrices of the V chain are reused
  GP_HIP(hipMemsetAsync(c->info, 0, 4 * sizeof(int), s));
  GP_TRY(launch_estep_proj_rows(a, lda, k, m, f, r, n, nrows, A, sv, u, s));
  GP_TRY(launch_estep_proj_scale(aL, ldal, k, n, nrows, sv, u, Y, want_moments ? aLp : nullptr, ld, npc, part, s));
  GP_TRY((launch_reduce_slices<double, double>(part, npc, nrows / 32, t2, npc, s)));   // t2 = (a L)^T u
  // W = I + Y^T Y  (= I + L^T G L, G = A^2 a^T diag(f) a), lower tiles, identity on the padding
  GP_TRY(gemm_full(s, 1, 1, npc, npc, nrows, 1.0, Y, ld, Y, ld, 0.0, c->Kbuf, ld, 1, 0, 0, 0, c->sk_ws[0]));
  GP_TRY(launch_add_diag(c->Kbuf, ld, npc, 1.0, s));
  CholBufs b{c->Kbuf, c->Lbuf, c->Libuf, c->Tmp, ld, c->info, 0, c->sk_ws[0]};
  GP_TRY(potrf_rec(b, 0, npc, 1, s));
  // m_new = L W^-1 (a L)^T u
  GP_TRY(launch_trmv_lower(c->Libuf, ld, npc, t2, z1, s));
  GP_TRY(launch_trmv_lower_t(c->Libuf, ld, npc, z1, z, c->trmv_part, s));
  GP_TRY(launch_pack_lower(L, ldl, k, Lp, ld, npc, s));
  GP_TRY(launch_trmv_lower(Lp, ld, npc, z, mo, s));
  // V_new = P P^T, P = L L_W^-T  (= (K~^-1 + G)^-1 = solve(I + K~ G, K~), utils.py:1430)
  GP_TRY(gemm_full(s, 0, 0, npc, npc, npc, 1.0, Lp, ld, c->Libuf, ld, 0.0, P, ld, 0, 1, 2, 0, c->sk_ws[0]));
  GP_TRY(gemm_full(s, 0, 0, npc, npc, npc, 1.0, P, ld, P, ld, 0.0, V, ld, 1, 0, 0, 0, c->sk_ws[0]));
  GP_HIP(hipMemcpyAsync(m_new, mo, (size_t)k * sizeof(double), hipMemcpyDeviceToDevice, s));
  GP_TRY(launch_unpack_sym(V, ld, k, V_new, ldv, s));   // symmetric by construction (utils.py:1438)
  if (want_moments) {
    // the moments of lambda the caller evaluates next (utils.py:1090, 1101), from Z = aL L_W^-T: a V_new a^T = Z Z^T
    GP_TRY(gemm_full(s, 0, 0, nrows, npc, npc, 1.0, aLp, ld, c->Libuf, ld, 0.0, Zm, ld, 0, 0, 2, 0, c->sk_ws[0]));
    GP_TRY(launch_estep_proj_moments(Zm, ld, k, z1, kv0, n, lam_m_out, lam_var_out, s));
  }
  GP_HIP(hipMemcpyAsync(c->info_host, c->info, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  GP_HIP(hipStreamSynchronize(s));
  if (c->info_host[0] != 0) {
    set_error("gpfit_estep_projected: I + L^T G L is not positive definite (NaN or negative firing rates?)");
    return c->info_host[0];
  }
  return 0;
}

int gpfit_fparam_eval(gpfit_ctx* c, void* stream, const double* lam_m, const double* lam_var, const double* r,
                      int64_t N, double logA, int closed_form_lambda0, double lambda0_in, double* f_out,
                      double* out_host) {
  if (!c || !lam_m || !lam_var || !r || !out_host || N <= 0) {
    set_error("gpfit_fparam_eval: bad argument");
    return -3;
  }
  GP_CTX_ENTER(c, "gpfit_fparam_eval");
  hipStream_t s = (hipStream_t)stream;
  // the seven results go straight to the context's pinned, device-mapped scalars (no copy command behind the kernel:
  // the L-BFGS of the rate parameters calls this ~6 times per E-step and waits for every answer)
  GP_TRY(launch_fparam(lam_m, lam_var, r, (int)N, std::exp(logA), closed_form_lambda0, lambda0_in, f_out,
                       c->scal_host + 32, s));
  GP_HIP(hipStreamSynchronize(s));
  for (int i = 0; i < 7; ++i) out_host[i] = c->scal_host[32 + i];
  return 0;
}

}  // extern "C"
