// Cholesky leaf: factor one 128 x 128 diagonal block in LDS and invert the factor.
//   in : A (lower triangle read), symmetric positive definite
//   out: L (lower Cholesky factor, strict upper part of the block zeroed) and Linv = L^-1
// The recursive blocked algorithm in chol.hip turns every triangular solve into an MFMA GEMM
// against these explicit 128 x 128 inverses (and their recursively assembled parents), so
// this kernel is the only non-GEMM step of the factorisation.
//
// One workgroup of 1024 threads (16 waves); the whole block lives in LDS (128 x 129 doubles,
// 129 KiB of the CU's 160 KiB).  Right-looking column Cholesky followed by an in-place
// column-by-column inversion of the lower factor (LAPACK dtrti2 order, last column first).
// info: LAPACK-style -- index (1-based, offset by info_base) of the first non-positive
// pivot is recorded with atomicCAS on *info (0 = none so far); the block is then completed
// with the offending pivot replaced by 1 so that no NaN/Inf propagates into later kernels.
#include "common.h"

namespace gpfit {

constexpr int LEAF = 128;
constexpr int LEAF_LD = 129;
constexpr int LEAF_THREADS = 1024;

__global__ __launch_bounds__(LEAF_THREADS) void chol_leaf_kernel(const double* __restrict__ A, int64_t lda,
                                                                 double* __restrict__ L, int64_t ldl,
                                                                 double* __restrict__ Linv, int64_t ldi,
                                                                 int* __restrict__ info, int info_base) {
  extern __shared__ __attribute__((aligned(16))) double sm[];  // [128][129]
  const int tid = threadIdx.x;
  const int ri = tid >> 3;   // row owned by this thread (0..127)
  const int cs = tid & 7;    // column slot: columns cs, cs+8, ...

  // load lower triangle, zero the strict upper part
  for (int e = tid; e < LEAF * LEAF; e += LEAF_THREADS) {
    const int i = e >> 7, j = e & 127;
    sm[i * LEAF_LD + j] = (j <= i) ? A[(int64_t)i * lda + j] : 0.0;
  }
  __syncthreads();

  // ---- right-looking Cholesky ----
  for (int k = 0; k < LEAF; ++k) {
    double akk = sm[k * LEAF_LD + k];
    if (!(akk > 0.0)) {  // uniform: every thread reads the same value
      if (tid == 0) atomicCAS(info, 0, info_base + k + 1);
      akk = 1.0;
    }
    const double d = sqrt(akk);
    const double lik = (ri > k) ? sm[ri * LEAF_LD + k] / d : 0.0;
    __syncthreads();  // everyone has read column k (and the pivot) before it is rewritten
    if (cs == 0) {
      if (ri > k) sm[ri * LEAF_LD + k] = lik;
      else if (ri == k) sm[k * LEAF_LD + k] = d;
    }
    __syncthreads();
    if (ri > k) {
#pragma unroll 4
      for (int j = k + 1 + cs; j <= ri; j += 8) sm[ri * LEAF_LD + j] -= lik * sm[j * LEAF_LD + k];
    }
    // next iteration's first read (pivot k+1, column k+1) is ordered by the barrier below
    __syncthreads();
  }

  // write L
  for (int e = tid; e < LEAF * LEAF; e += LEAF_THREADS) {
    const int i = e >> 7, j = e & 127;
    L[(int64_t)i * ldl + j] = sm[i * LEAF_LD + j];
  }
  __syncthreads();

  // ---- in-place inverse of the lower factor, last column first ----
  // column j:  x_jj = 1/l_jj ;  x[j+1:, j] = -(T * l[j+1:, j]) * x_jj  with T = inverse of
  // the trailing block (already in place, lower triangular).
  for (int j = LEAF - 1; j >= 0; --j) {
    const double xjj = 1.0 / sm[j * LEAF_LD + j];
    double s = 0.0;
    if (ri > j) {
      for (int k = j + 1 + cs; k <= ri; k += 8) s += sm[ri * LEAF_LD + k] * sm[k * LEAF_LD + j];
    }
    // reduce the 8 partial sums of a row (8 consecutive lanes)
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    __syncthreads();  // all reads of old column j done
    if (cs == 0) {
      if (ri > j) sm[ri * LEAF_LD + j] = -s * xjj;
      else if (ri == j) sm[j * LEAF_LD + j] = xjj;
    }
    __syncthreads();
  }
  for (int e = tid; e < LEAF * LEAF; e += LEAF_THREADS) {
    const int i = e >> 7, j = e & 127;
    Linv[(int64_t)i * ldi + j] = sm[i * LEAF_LD + j];
  }
}

int launch_chol_leaf(const double* A, int64_t lda, double* L, int64_t ldl, double* Linv, int64_t ldi,
                     int* info, int info_base, hipStream_t s) {
  static bool attr_set = false;
  const size_t lds = sizeof(double) * LEAF * LEAF_LD;
  if (!attr_set) {
    GP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_leaf_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(chol_leaf_kernel, dim3(1), dim3(LEAF_THREADS), lds, s, A, lda, L, ldl, Linv, ldi, info,
                     info_base);
  GP_HIP(hipGetLastError());
  return 0;
}

}  // namespace gpfit
