// Cholesky leaf: factor one 128 x 128 diagonal block and invert the factor, entirely in LDS.
//   in : A (lower triangle read), symmetric positive definite
//   out: L (lower Cholesky factor, strict upper part of the block zeroed) and Linv = L^-1
// The recursive blocked algorithm in fit.hip turns every triangular solve into an MFMA GEMM
// against these explicit inverses (and their recursively assembled parents), so this kernel
// is the only non-GEMM step of the factorisation and sits on its critical path 2N/128 times
// per fit: it is written for latency.
//
// One workgroup of 4 waves; the block lives in LDS as [128][130] doubles (130: the MFMA
// fragment reads of 16 rows x 4 k's then hit 64 distinct banks).  Blocked right-looking
// factorisation over 16-column panels:
//   (1) the 16 x 16 diagonal block is factored by ONE wave with a row per lane in registers;
//       pivots and multipliers move between lanes with v_readlane (compile-time lane ids,
//       fully unrolled) -- "wavefront shuffles" instead of LDS round trips or barriers;
//   (2) the rows below solve against it by forward substitution, one row per lane, the
//       16 x 16 factor read from LDS as wave-uniform broadcasts;
//   (3) the trailing 16 x 16 tiles take their rank-16 update on v_mfma_f64_16x16x4_f64.
// Then L^-1 is assembled in place: 16 x 16 diagonal inverses (a column per lane), followed by
// three MFMA merge levels X21 = -X22 (L21 X11) for s = 16, 32, 64.
// info: LAPACK-style -- 1-based index (offset by info_base) of the first non-positive pivot is
// recorded with atomicCAS on *info (0 = none so far); the block is then completed with the
// offending pivot replaced by 1 so that no NaN/Inf propagates into later kernels.
#include "common.h"
#include <cstdlib>

namespace gpfit {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int LEAF = 128;
constexpr int LLD = 130;
constexpr int LEAF_THREADS = 256;
constexpr size_t LEAF_LDS_BYTES = sizeof(double) * (LEAF * LLD + LEAF);

__device__ __forceinline__ double readlane_d(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void tri_decode(int t, int& a, int& b) {  // t -> (a, b), b <= a
  int i = 0;
  while ((i + 1) * (i + 2) / 2 <= t) ++i;
  a = i;
  b = t - i * (i + 1) / 2;
}

__global__ __launch_bounds__(LEAF_THREADS) void chol_leaf_kernel(const double* __restrict__ A, int64_t lda,
                                                                 double* __restrict__ L, int64_t ldl,
                                                                 double* __restrict__ Linv, int64_t ldi,
                                                                 int* __restrict__ info, int info_base, int dbg) {
  extern __shared__ __attribute__((aligned(16))) double S[];  // [128][130] + rdiag[128]
  double* rdiag = S + LEAF * LLD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;

  // the block is read with all 32 row segments of a thread in flight at once (a rolled loop
  // would pay one L2/HBM round trip per row: ~20 us of pure latency on the critical path)
  {
    double2 v[32];
#pragma unroll
    for (int it = 0; it < 32; ++it) {
      const int e = tid + it * LEAF_THREADS;
      const int i = e >> 6, j = (e & 63) * 2;
      v[it] = (j <= i) ? *reinterpret_cast<const double2*>(A + (int64_t)i * lda + j) : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int it = 0; it < 32; ++it) {
      const int e = tid + it * LEAF_THREADS;
      const int i = e >> 6, j = (e & 63) * 2;
      if (j + 1 > i) v[it].y = 0.0;
      *reinterpret_cast<double2*>(S + i * LLD + j) = v[it];
    }
  }
  __syncthreads();

  // ======================= blocked Cholesky over 16-column panels =======================
  for (int kb = 0; kb < 8; ++kb) {
    const int c0 = 16 * kb;
    // ---- (1) diagonal 16 x 16 block: one wave, one row per lane, readlane broadcasts
    if (wave == 0 && !(dbg & 1)) {
      double v[16], rk[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = (lane < 16) ? S[(c0 + lane) * LLD + c0 + j] : 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        double p = readlane_d(v[k], k);
        if (!(p > 0.0)) {  // wave-uniform
          if (lane == 0) atomicCAS(info, 0, info_base + c0 + k + 1);
          p = 1.0;
        }
        // 1/sqrt(p) from v_rsq_f64 + two Newton steps, sqrt(p) = p*y with one Heron correction:
        // ~1/3 of the dependent latency of sqrt() followed by a division, same last-bit quality
        double rinv = __builtin_amdgcn_rsq(p);
        const double hp = 0.5 * p;
        rinv = rinv * fma(-hp * rinv, rinv, 1.5);
        rinv = rinv * fma(-hp * rinv, rinv, 1.5);
        double dkk = p * rinv;
        dkk = fma(fma(-dkk, dkk, p), 0.5 * rinv, dkk);
        rk[k] = rinv;
        v[k] = (lane == k) ? dkk : v[k] * rinv;
#pragma unroll
        for (int j = k + 1; j < 16; ++j) {
          const double ljk = readlane_d(v[k], j);
          v[j] -= v[k] * ljk;
        }
      }
      if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) S[(c0 + lane) * LLD + c0 + j] = (j <= lane) ? v[j] : 0.0;
      }
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) rdiag[c0 + k] = rk[k];
      }
    }
    __syncthreads();
    // ---- (2) rows below the diagonal block: X * Ld^T = P, one row per lane
    const int m = LEAF - c0 - 16;
    if (tid < m && !(dbg & 2)) {
      const int row = c0 + 16 + tid;
      double x[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) x[j] = S[row * LLD + c0 + j];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        double acc = x[k];
#pragma unroll
        for (int j = 0; j < k; ++j) acc -= x[j] * S[(c0 + k) * LLD + c0 + j];
        x[k] = acc * rdiag[c0 + k];
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) S[row * LLD + c0 + j] = x[j];
    }
    __syncthreads();
    // ---- (3) trailing rank-16 update of the 16 x 16 tiles (ta >= tb) on MFMA
    const int nt = 7 - kb, ntiles = nt * (nt + 1) / 2;
    for (int t = wave; t < ((dbg & 4) ? 0 : ntiles); t += 4) {
      int ta, tb;
      tri_decode(t, ta, tb);
      const int i0 = c0 + 16 + 16 * ta, j0 = c0 + 16 + 16 * tb;
      v4d acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = S[(i0 + fq + 4 * r) * LLD + j0 + fr];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const double a = -S[(i0 + fr) * LLD + c0 + 4 * kk + fq];
        const double b = S[(j0 + fr) * LLD + c0 + 4 * kk + fq];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) S[(i0 + fq + 4 * r) * LLD + j0 + fr] = acc[r];
    }
    __syncthreads();
  }

  for (int e = tid; e < LEAF * LEAF / 2; e += LEAF_THREADS) {
    const int i = e >> 6, j = (e & 63) * 2;
    double2 v = *reinterpret_cast<const double2*>(S + i * LLD + j);
    if (j > i) v.x = 0.0;
    if (j + 1 > i) v.y = 0.0;
    *reinterpret_cast<double2*>(L + (int64_t)i * ldl + j) = v;
  }
  __syncthreads();

  // ======================= in-place inverse of the lower factor =======================
  // (I1) the eight 16 x 16 diagonal blocks: forward substitution on the identity, one column
  //      per lane (lanes 0-15: block 2*wave, lanes 16-31: block 2*wave+1)
  if (dbg & 32) return;
  if (lane < 32 && !(dbg & 8)) {
    const int b0 = 16 * (2 * wave + (lane >> 4));
    const int j = lane & 15;
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      double acc = (i == j) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < i; ++k) acc -= S[(b0 + i) * LLD + b0 + k] * x[k];
      x[i] = acc * rdiag[b0 + i];
    }
    // all reads of the block happen above (the wave runs in lockstep), writes below
#pragma unroll
    for (int i = 0; i < 16; ++i) S[(b0 + i) * LLD + b0 + j] = x[i];
  }
  __syncthreads();

  // (I2) merge levels: for each pair (X11, X22) of inverted s x s diagonal blocks at offset o,
  //      X21 = -X22 * (L21 * X11), 16 x 16 tiles on MFMA, product kept in registers between the
  //      two passes so the update is in place.
  for (int s = 16; s <= ((dbg & 16) ? 0 : 64); s *= 2) {
    const int tps = s / 16;                  // tiles per side of one X21 block
    const int tiles_per_merge = tps * tps;
    const int total = (LEAF / (2 * s)) * tiles_per_merge;  // 4, 8, 16
    v4d acc[4];
    // pass a: tmp = L21 * X11   (X11 lower: k-blocks >= tb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = wave + 4 * q;
      acc[q] = v4d{0.0, 0.0, 0.0, 0.0};
      if (t < total) {
        const int mg = t / tiles_per_merge, tt = t % tiles_per_merge;
        const int ta = tt / tps, tb = tt % tps;
        const int o = mg * 2 * s;
        const int i0 = o + s + 16 * ta, j0 = o + 16 * tb;
        for (int kblk = tb; kblk < tps; ++kblk) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int k = o + 16 * kblk + 4 * kk + fq;
            const double a = S[(i0 + fr) * LLD + k];
            const double b = S[k * LLD + j0 + fr];
            acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = wave + 4 * q;
      if (t < total) {
        const int mg = t / tiles_per_merge, tt = t % tiles_per_merge;
        const int ta = tt / tps, tb = tt % tps;
        const int o = mg * 2 * s;
        const int i0 = o + s + 16 * ta, j0 = o + 16 * tb;
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(i0 + fq + 4 * r) * LLD + j0 + fr] = acc[q][r];
      }
    }
    __syncthreads();
    // pass b: X21 = -X22 * tmp   (X22 lower: k-blocks <= ta)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = wave + 4 * q;
      acc[q] = v4d{0.0, 0.0, 0.0, 0.0};
      if (t < total) {
        const int mg = t / tiles_per_merge, tt = t % tiles_per_merge;
        const int ta = tt / tps, tb = tt % tps;
        const int o = mg * 2 * s;
        const int i0 = o + s + 16 * ta, j0 = o + 16 * tb;
        for (int kblk = 0; kblk <= ta; ++kblk) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int k = o + s + 16 * kblk + 4 * kk + fq;
            const double a = -S[(i0 + fr) * LLD + k];
            const double b = S[k * LLD + j0 + fr];
            acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = wave + 4 * q;
      if (t < total) {
        const int mg = t / tiles_per_merge, tt = t % tiles_per_merge;
        const int ta = tt / tps, tb = tt % tps;
        const int o = mg * 2 * s;
        const int i0 = o + s + 16 * ta, j0 = o + 16 * tb;
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(i0 + fq + 4 * r) * LLD + j0 + fr] = acc[q][r];
      }
    }
    __syncthreads();
  }

  for (int e = tid; e < LEAF * LEAF / 2; e += LEAF_THREADS) {
    const int i = e >> 6, j = (e & 63) * 2;
    double2 v = *reinterpret_cast<const double2*>(S + i * LLD + j);
    if (j > i) v.x = 0.0;
    if (j + 1 > i) v.y = 0.0;
    *reinterpret_cast<double2*>(Linv + (int64_t)i * ldi + j) = v;
  }
}

int launch_chol_leaf(const double* A, int64_t lda, double* L, int64_t ldl, double* Linv, int64_t ldi,
                     int* info, int info_base, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    GP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_leaf_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)LEAF_LDS_BYTES));
    attr_set = true;
  }
  static int dbg = -1;
  if (dbg < 0) dbg = getenv("GPFIT_LEAF_DBG") ? atoi(getenv("GPFIT_LEAF_DBG")) : 0;
  hipLaunchKernelGGL(chol_leaf_kernel, dim3(1), dim3(LEAF_THREADS), LEAF_LDS_BYTES, s, A, lda, L, ldl, Linv, ldi,
                     info, info_base, dbg);
  GP_HIP(hipGetLastError());
  return 0;
}

}  // namespace gpfit
