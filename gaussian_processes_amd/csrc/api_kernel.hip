// C-ABI entry points that materialise kernel objects for callers of the reference's public
// functions: localker (C, dC), acosker (K and the six dK, square / rectangular / diagonal).
// The fused fit path (fit.hip) never materialises dK; these exist so that the drop-in module
// can serve notebooks that call utils.acosker / utils.localker directly.
#include "context.h"
#include "gpfit_mi355x.h"

#include <cmath>

using namespace gpfit;

#define GP_TRY(expr)            \
  do {                          \
    int _rc = (expr);           \
    if (_rc != 0) return _rc;   \
  } while (0)

static int gemm_kk(hipStream_t s, int M, int N, int K, const double* A, int64_t lda, const double* B, int64_t ldb,
                   double* C, int64_t ldc) {
  GemmArgs g{};
  g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K; g.alpha = 1.0; g.beta = 0.0; g.a_kmajor = 1; g.b_kmajor = 1;
  g.batch = 1; g.split_k = 1;
  return launch_gemm(g, s);
}

extern "C" {

int gpfit_localker(gpfit_ctx* c, void* stream, const double* theta, int n_rows, int n_cols,
                   const uint8_t* mask_host, int64_t d, double* C_dev, double* dC_dev) {
  if (!c || !theta || !mask_host || !C_dev || d <= 0 || n_rows * n_cols > c->dfull_cap) {
    set_error("gpfit_localker: bad argument or image larger than the context capacity");
    return -3;
  }
  GP_CTX_ENTER(c, "gpfit_localker");
  hipStream_t s = (hipStream_t)stream;
  int k = 0;
  for (int p = 0; p < n_rows * n_cols; ++p)
    if (mask_host[p]) c->pix_host[k++] = p;
  if (k != d) {
    set_error("gpfit_localker: d does not match the mask");
    return -3;
  }
  Theta th;
  th.sigma0 = theta[0]; th.eps0x = theta[1]; th.eps0y = theta[2]; th.logbeta = theta[3]; th.logrho = theta[4];
  th.amp = theta[5]; th.eb = std::exp(theta[3]); th.er = std::exp(theta[4]);
  GP_HIP(hipMemcpyAsync(c->pix, c->pix_host, (size_t)d * sizeof(int), hipMemcpyHostToDevice, s));
  GP_TRY(launch_localker(th, c->pix, (int)d, (int)d, n_rows, n_cols, C_dev, d, dC_dev, s));
  GP_HIP(hipStreamSynchronize(s));  // pix_host may be reused by the next call
  return 0;
}

// order of the five dC matrices on input: Amp, -2log2beta, -log2rho2, eps_0x, eps_0y
// order of the six dK matrices on output: theta dict order (sigma_0, eps_0x, eps_0y, -2log2beta, -log2rho2, Amp)
static const int kDcToTheta[5] = {5, 3, 4, 1, 2};

int gpfit_acosker(gpfit_ctx* c, void* stream, double sigma0, const double* x1, int64_t ld1, int64_t n1,
                  const double* x2, int64_t ld2, int64_t n2, int64_t d, const double* C, int64_t ldC,
                  const double* dC, double* K, int64_t ldk, double* dK) {
  if (!c || !x1 || !x2 || !C || !K || n1 <= 0 || n2 <= 0 || d <= 0) {
    set_error("gpfit_acosker: bad argument");
    return -3;
  }
  GP_CTX_ENTER(c, "gpfit_acosker");
  hipStream_t s = (hipStream_t)stream;
  const int dp = (int)round_up(d, 32), np1 = (int)round_up(n1, TILE), np2 = (int)round_up(n2, TILE);
  if (dp > c->dp_cap || np1 > c->np_cap || np2 > c->np_cap) {
    set_error("gpfit_acosker: problem larger than the context capacity");
    return -3;
  }
  const bool same = (x1 == x2 && n1 == n2 && ld1 == ld2);
  const bool square = (n1 == n2);
  const double s0sq = sigma0 * sigma0;
  GP_TRY(launch_pad_copy(C, ldC, (int)d, (int)d, c->Cmat, dp, dp, dp, s));
  GP_TRY(launch_gather<double>(x1, ld1, (int)n1, nullptr, (int)d, dp, np1, c->Xt, np1, nullptr, 0, s));
  GP_TRY(gemm_kk(s, dp, np1, dp, c->Cmat, dp, c->Xt, np1, c->XCt, np1));
  GP_TRY(launch_qvec(c->Xt, c->XCt, np1, dp, (int)n1, np1, s0sq, c->Kvec, c->q, s));
  const double* Xt2 = c->Xt;
  const double* q2 = c->q;
  if (!same) {
    GP_TRY(launch_gather<double>(x2, ld2, (int)n2, nullptr, (int)d, dp, np2, c->Xt2, np2, nullptr, 0, s));
    GP_TRY(gemm_kk(s, dp, np2, dp, c->Cmat, dp, c->Xt2, np2, c->XCt2, np2));
    GP_TRY(launch_qvec(c->Xt2, c->XCt2, np2, dp, (int)n2, np2, s0sq, c->hvec, c->q2, s));
    Xt2 = c->Xt2;
    q2 = c->q2;
  }
  const bool lower = same && !dK;
  {
    GramArgs g{};
    g.XCt = c->XCt; g.Xt = Xt2; g.q1 = c->q; g.q2 = q2; g.Kout = K; g.Cos = dK ? c->Cos : nullptr;
    g.ld1 = np1; g.ld2 = np2; g.ldk = ldk; g.np1 = np1; g.np2 = np2; g.nv1 = (int)n1; g.nv2 = (int)n2; g.Kd = dp;
    g.s0sq = s0sq; g.lower = lower ? 1 : 0; g.pad_identity = 0;
    g.ldcos = np2;
    GP_TRY(launch_gram(g, s));
  }
  if (square) {
    if (lower) GP_TRY(launch_symmetrize(K, ldk, (int)n1, s));
    else GP_TRY(launch_symmetrize_avg(K, ldk, (int)n1, s));  // utils.py:1024-1025
  }
  if (dK) {
    const int64_t nn = n1 * n2;
    GP_TRY(launch_dk_sigma0(c->Cos, np2, c->q, q2, (int)n1, (int)n2, sigma0, dK, n2, s));
    if (dC) {
      for (int p = 0; p < 5; ++p) {
        double* out = dK + (int64_t)kDcToTheta[p] * nn;
        GP_TRY(launch_pad_copy(dC + (int64_t)p * d * d, d, (int)d, (int)d, c->dCpad, dp, dp, dp, s));
        GP_TRY(gemm_kk(s, dp, np1, dp, c->dCpad, dp, c->Xt, np1, c->XDt, np1));
        GP_TRY(launch_dq(c->Xt, c->XDt, np1, dp, (int)n1, c->q, c->dq1, nullptr, s));
        const double* dq2 = c->dq1;
        if (!same) {
          GP_TRY(gemm_kk(s, dp, np2, dp, c->dCpad, dp, c->Xt2, np2, c->XDt2, np2));
          GP_TRY(launch_dq(c->Xt2, c->XDt2, np2, dp, (int)n2, q2, c->dq2, nullptr, s));
          dq2 = c->dq2;
        }
        // H = x1 dC_p x2^T straight into the caller's dK_p, then the element-wise chain in place
        GP_TRY(gemm_kk(s, (int)n1, (int)n2, dp, c->XDt, np1, Xt2, np2, out, n2));
        GP_TRY(launch_dk_metric(out, n2, c->Cos, np2, c->q, q2, c->dq1, dq2, (int)n1, (int)n2, s));
      }
    }
  }
  return 0;
}

int gpfit_acosker_pullback(gpfit_ctx* c, void* stream, double sigma0, const double* x1, int64_t ld1, int64_t n1,
                           const double* x2, int64_t ld2, int64_t n2, int64_t d, const double* C, int64_t ldC,
                           const double* W, int64_t ldw, const double* t1_extra, double* M_out, int64_t ldm,
                           double* out_host) {
  if (!c || !x1 || !x2 || !C || !W || !M_out || !out_host || n1 <= 0 || n2 <= 0 || d <= 0) {
    set_error("gpfit_acosker_pullback: bad argument");
    return -3;
  }
  GP_CTX_ENTER(c, "gpfit_acosker_pullback");
  hipStream_t s = (hipStream_t)stream;
  const int dp = (int)round_up(d, 32), np1 = (int)round_up(n1, TILE), np2 = (int)round_up(n2, TILE);
  if (dp > c->dp_cap || np1 > c->np_cap || np2 > c->np_cap) {
    set_error("gpfit_acosker_pullback: problem larger than the context capacity");
    return -3;
  }
  const double s0sq = sigma0 * sigma0;
  double* X1m = c->Xm;     // [np1][dp] row-major copies of the operands
  double* X2m = c->XDt;    // the [dp][np] scratch matrices have the same element count
  double* Zm = c->XDt2;
  c->lv_valid = false; c->lv32_valid = false;
  // forward pieces: q1, q2, the cosine matrix
  GP_TRY(launch_pad_copy(C, ldC, (int)d, (int)d, c->Cmat, dp, dp, dp, s));
  GP_TRY(launch_gather<double>(x1, ld1, (int)n1, nullptr, (int)d, dp, np1, c->Xt, np1, X1m, dp, s));
  GP_TRY(gemm_kk(s, dp, np1, dp, c->Cmat, dp, c->Xt, np1, c->XCt, np1));
  GP_TRY(launch_qvec(c->Xt, c->XCt, np1, dp, (int)n1, np1, s0sq, c->Kvec, c->q, s));
  GP_TRY(launch_gather<double>(x2, ld2, (int)n2, nullptr, (int)d, dp, np2, c->Xt2, np2, X2m, dp, s));
  GP_TRY(gemm_kk(s, dp, np2, dp, c->Cmat, dp, c->Xt2, np2, c->XCt2, np2));
  GP_TRY(launch_qvec(c->Xt2, c->XCt2, np2, dp, (int)n2, np2, s0sq, c->hvec, c->q2, s));
  {
    GramArgs g{};
    g.XCt = c->XCt; g.Xt = c->Xt2; g.q1 = c->q; g.q2 = c->q2; g.Kout = c->Kbuf; g.Cos = c->Cos;
    g.ld1 = np1; g.ld2 = np2; g.ldk = np2; g.np1 = np1; g.np2 = np2; g.nv1 = (int)n1; g.nv2 = (int)n2; g.Kd = dp;
    g.s0sq = s0sq; g.lower = 0; g.pad_identity = 0;
    g.ldcos = np2;
    GP_TRY(launch_gram(g, s));
  }
  // adjoint pass: A_w [np1][np2], t1 = u1 / (2 q1) + extra, t2 = u2 / (2 q2), the three sums
  double* t1 = c->tvec;
  double* t2 = c->tvec + c->np_cap;
  GP_TRY(launch_adjoint_rect(W, ldw, c->Cos, np2, c->q, c->q2, (int)n1, (int)n2, np1, np2, c->Abuf, np2, c->upart,
                             c->vpart, c->rect_part, t1_extra, t1, t2, c->rpad, c->mpad, c->scal + 20, s));
  // M = x1^T (A_w x2 + t1 o x1) + x2^T (t2 o x2)
  {
    GemmArgs g{};
    g.A = c->Abuf; g.B = X2m; g.C = c->Ybuf; g.lda = np2; g.ldb = dp; g.ldc = dp;
    g.M = np1; g.N = dp; g.K = np2; g.alpha = 1.0; g.beta = 0.0; g.a_kmajor = 0; g.b_kmajor = 1; g.batch = 1; g.split_k = 1;
    GP_TRY(launch_gemm(g, s));
  }
  GP_TRY(launch_rowscale_add(c->Ybuf, dp, X1m, dp, t1, np1, dp, s));
  GP_HIP(hipMemsetAsync(Zm, 0, (size_t)np2 * dp * sizeof(double), s));
  GP_TRY(launch_rowscale_add(Zm, dp, X2m, dp, t2, np2, dp, s));
  auto xty = [&](const double* Xa, const double* Yb, int np, double* out) -> int {
    GemmArgs g{};
    g.A = Xa; g.B = Yb; g.C = c->Mpart; g.lda = dp; g.ldb = dp; g.ldc = dp;
    g.M = dp; g.N = dp; g.K = np; g.alpha = 1.0; g.beta = 0.0; g.a_kmajor = 1; g.b_kmajor = 1;
    g.batch = 1; g.split_k = c->split_k_M; g.sC = (int64_t)dp * dp;
    GP_TRY(launch_gemm(g, s));
    return launch_reduce_slices(c->Mpart, (int64_t)dp * dp, c->split_k_M, out, (int64_t)dp * dp, s);
  };
  GP_TRY(xty(X1m, c->Ybuf, np1, c->Mmat));
  GP_TRY(xty(X2m, Zm, np2, c->dCpad));
  GP_TRY(launch_axpby_block<double>(c->Mmat, dp, c->dCpad, dp, dp, dp, 1.0, 1.0, s));
  GP_HIP(hipMemcpy2DAsync(M_out, (size_t)ldm * sizeof(double), c->Mmat, (size_t)dp * sizeof(double),
                          (size_t)d * sizeof(double), (size_t)d, hipMemcpyDeviceToDevice, s));
  GP_HIP(hipMemcpyAsync(c->scal_host, c->scal, 64 * sizeof(double), hipMemcpyDeviceToHost, s));
  GP_HIP(hipStreamSynchronize(s));
  out_host[0] = c->scal_host[20];
  out_host[1] = c->scal_host[21];
  out_host[2] = c->scal_host[22];
  return 0;
}

int gpfit_acosker_diag(gpfit_ctx* c, void* stream, double sigma0, const double* x1, int64_t ld1, int64_t n1,
                       int64_t d, const double* C, int64_t ldC, const double* dC, double* Kvec, double* dKvec) {
  if (!c || !x1 || !C || !Kvec || n1 <= 0 || d <= 0) {
    set_error("gpfit_acosker_diag: bad argument");
    return -3;
  }
  GP_CTX_ENTER(c, "gpfit_acosker_diag");
  hipStream_t s = (hipStream_t)stream;
  const int dp = (int)round_up(d, 32), np1 = (int)round_up(n1, TILE);
  if (dp > c->dp_cap || np1 > c->np_cap) {
    set_error("gpfit_acosker_diag: problem larger than the context capacity");
    return -3;
  }
  const double s0sq = sigma0 * sigma0;
  GP_TRY(launch_pad_copy(C, ldC, (int)d, (int)d, c->Cmat, dp, dp, dp, s));
  GP_TRY(launch_gather<double>(x1, ld1, (int)n1, nullptr, (int)d, dp, np1, c->Xt, np1, nullptr, 0, s));
  GP_TRY(gemm_kk(s, dp, np1, dp, c->Cmat, dp, c->Xt, np1, c->XCt, np1));
  GP_TRY(launch_qvec(c->Xt, c->XCt, np1, dp, (int)n1, np1, s0sq, c->Kvec, c->q, s));
  GP_HIP(hipMemcpyAsync(Kvec, c->Kvec, (size_t)n1 * sizeof(double), hipMemcpyDeviceToDevice, s));  // utils.py:1029
  if (dKvec) {
    GP_TRY(launch_fill(dKvec, n1, 2.0 * s0sq / sigma0, s));  // utils.py:1036
    if (dC) {
      for (int p = 0; p < 5; ++p) {
        GP_TRY(launch_pad_copy(dC + (int64_t)p * d * d, d, (int)d, (int)d, c->dCpad, dp, dp, dp, s));
        GP_TRY(gemm_kk(s, dp, np1, dp, c->dCpad, dp, c->Xt, np1, c->XDt, np1));
        GP_TRY(launch_dq(c->Xt, c->XDt, np1, dp, (int)n1, c->q, nullptr, dKvec + (int64_t)kDcToTheta[p] * n1, s));  // :1042
      }
    }
  }
  return 0;
}

}  // extern "C"
