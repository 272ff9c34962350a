// C-ABI entry points: error reporting and the raw fp64 MFMA GEMM primitive.
#include "common.h"
#include "gpfit_mi355x.h"

namespace gpfit {
static thread_local std::string g_err;
void set_error(const std::string& m) { g_err = m; }
}  // namespace gpfit

extern "C" {

int gpfit_version(void) { return GPFIT_VERSION; }

const char* gpfit_last_error(void) { return gpfit::g_err.c_str(); }

int gpfit_dgemm_ex(void* stream, int a_kmajor, int b_kmajor, int M, int N, int K, double alpha,
                   const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* C,
                   int64_t ldc, int out_lower, int a_tri, int b_tri, int walk, int tile);

int gpfit_dgemm(void* stream, int a_kmajor, int b_kmajor, int M, int N, int K, double alpha,
                const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* C,
                int64_t ldc, int out_lower, int a_tri, int b_tri) {
  return gpfit_dgemm_ex(stream, a_kmajor, b_kmajor, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, out_lower,
                        a_tri, b_tri, 0, 0);
}

int gpfit_dgemm_ex(void* stream, int a_kmajor, int b_kmajor, int M, int N, int K, double alpha,
                   const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* C,
                   int64_t ldc, int out_lower, int a_tri, int b_tri, int walk, int tile) {
  if ((M & 1) || (N & 1)) {
    gpfit::set_error("gpfit_dgemm: M and N must be even");
    return -3;
  }
  gpfit::GemmArgs g{};
  g.A = A; g.B = B; g.C = C;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K;
  g.alpha = alpha; g.beta = beta;
  g.a_kmajor = a_kmajor; g.b_kmajor = b_kmajor;
  g.out_lower = out_lower; g.a_tri = a_tri; g.b_tri = b_tri;
  g.batch = 1; g.split_k = 1; g.reverse = walk & 15; g.tile = tile;
  g.half_occ = (walk >> 4) & 1;  // bit 4: one workgroup per CU (a long GEMM that must leave room for latency-bound kernels)
  return gpfit::launch_gemm(g, (hipStream_t)stream);
}

}  // extern "C"
