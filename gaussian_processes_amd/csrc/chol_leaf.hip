// Cholesky leaf: factor one 128 x 128 diagonal block and invert the factor, entirely in LDS.
//   in : A (lower triangle read), symmetric positive definite
//   out: L (lower Cholesky factor, strict upper part of the block zeroed) and Linv = L^-1
// The recursive blocked algorithm in fit.hip turns every triangular solve into an MFMA GEMM
// against these explicit inverses (and their recursively assembled parents), so this kernel
// is the only non-GEMM step of the factorisation and sits on its critical path 2N/128 times
// per fit: it is written for latency.
//
// One workgroup of 4 waves; the block lives in LDS as [128][130] elements (130: the MFMA
// fragment reads of 16 rows x 4 k's then spread over the banks).  Blocked right-looking
// factorisation over 16-column panels:
//   (1) the 16 x 16 diagonal block is factored by ONE wave with a row per lane in registers;
//       pivots and multipliers move between lanes with v_readlane (compile-time lane ids,
//       fully unrolled) -- "wavefront shuffles" instead of LDS round trips or barriers;
//   (2) the rows below solve against it by forward substitution, one row per lane, the
//       16 x 16 factor read from LDS as wave-uniform broadcasts;
//   (3) the trailing 16 x 16 tiles take their rank-16 update on the 16x16x4 MFMA -- with a
//       look-ahead: only the first tile column (next diagonal block + next panel) is updated
//       before the next diagonal factor starts; the other tiles are done by waves 1-3 WHILE
//       wave 0 factors the next diagonal block (66 -> 56 us per leaf).
// Then L^-1 is assembled in place: 16 x 16 diagonal inverses (a column per lane), followed by
// three MFMA merge levels X21 = -X22 (L21 X11) for s = 16, 32, 64.
// info: LAPACK-style -- 1-based index (offset by info_base) of the first non-positive pivot is
// recorded with atomicCAS on *info (0 = none so far) once at the end (the pivot loop itself is
// branch-free); the block is completed with the offending pivot replaced by 1 so that no NaN/Inf
// propagates into later kernels.
// Instantiated for fp64 and fp32.
#include "gemm_core.h"
#include "kernels.h"

#include <atomic>
#include <cstdlib>

namespace gpfit {

constexpr int LEAF = 128;
constexpr int LLD = 130;
constexpr int LEAF_THREADS = 256;

__device__ __forceinline__ double readlane_r(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float readlane_r(float v, int src_lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// 1/sqrt(p) from the hardware estimate + two Newton steps, sqrt(p) = p*y with one Heron
// correction: ~1/3 of the dependent latency of sqrt() followed by a division.
__device__ __forceinline__ void rsqrt_sqrt(double p, double& rinv, double& root) {
  double y = __builtin_amdgcn_rsq(p);
  const double hp = 0.5 * p;
  y = y * fma(-hp * y, y, 1.5);
  y = y * fma(-hp * y, y, 1.5);
  double d = p * y;
  d = fma(fma(-d, d, p), 0.5 * y, d);
  rinv = y;
  root = d;
}
__device__ __forceinline__ void rsqrt_sqrt(float p, float& rinv, float& root) {
  float y = __builtin_amdgcn_rsqf(p);
  const float hp = 0.5f * p;
  y = y * fmaf(-hp * y, y, 1.5f);
  float d = p * y;
  d = fmaf(fmaf(-d, d, p), 0.5f * y, d);
  rinv = y;
  root = d;
}

__device__ __forceinline__ void tri_decode(int t, int& a, int& b) {  // t -> (a, b), b <= a
  int i = 0;
  while ((i + 1) * (i + 2) / 2 <= t) ++i;
  a = i;
  b = t - i * (i + 1) / 2;
}

template <typename R>
__global__ __launch_bounds__(LEAF_THREADS) void chol_leaf_kernel(const R* __restrict__ A, int64_t lda,
                                                                 R* __restrict__ L, int64_t ldl,
                                                                 R* __restrict__ Linv, int64_t ldi,
                                                                 int* __restrict__ info, int info_base, int dbg) {
  using V = typename Real<R>::vec_t;
  using Acc = typename Real<R>::acc_t;
  constexpr int EPC = Real<R>::EPC, CPR = LEAF / EPC, NCH = LEAF * CPR / LEAF_THREADS;
  extern __shared__ __attribute__((aligned(16))) unsigned char S_raw[];
  R* S = reinterpret_cast<R*>(S_raw);  // [128][130] + rdiag[128]
  R* rdiag = S + LEAF * LLD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;

  // the block is read with every row segment of a thread in flight at once (a rolled loop would
  // pay one L2/HBM round trip per row on the critical path)
  {
    V v[NCH];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int e = tid + it * LEAF_THREADS;
      const int i = e / CPR, j = (e % CPR) * EPC;
      v[it] = (j <= i) ? *reinterpret_cast<const V*>(A + (int64_t)i * lda + j) : V{};
    }
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int e = tid + it * LEAF_THREADS;
      const int i = e / CPR, j = (e % CPR) * EPC;
#pragma unroll
      for (int q = 0; q < EPC; ++q) S[i * LLD + j + q] = (j + q <= i) ? v[it][q] : (R)0;
    }
  }
  __syncthreads();

  // ======================= blocked Cholesky over 16-column panels =======================
  // rank-16 update of one 16 x 16 tile (i0, j0) of the trailing matrix with panel columns c0..c0+15
  auto trailing_tile = [&](int i0, int j0, int c0) {
    Acc acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = S[(i0 + Real<R>::crow(lane, r)) * LLD + j0 + fr];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const R a = -S[(i0 + fr) * LLD + c0 + 4 * kk + fq];
      const R b = S[(j0 + fr) * LLD + c0 + 4 * kk + fq];
      acc = Real<R>::mfma(a, b, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) S[(i0 + Real<R>::crow(lane, r)) * LLD + j0 + fr] = acc[r];
  };
  int first_bad = 0;  // wave 0: 1-based index of the first non-positive pivot of this block
  for (int kb = 0; kb < 8; ++kb) {
    const int c0 = 16 * kb;
    // ---- (1) diagonal 16 x 16 block: one wave, one row per lane, readlane broadcasts.
    //      Waves 1-3 meanwhile finish the trailing update of the PREVIOUS panel: its tiles right of
    //      the first tile column touch neither this diagonal block nor this panel (look-ahead).
    if (wave == 0 && !(dbg & 1)) {
      R v[16], rk[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = (lane < 16) ? S[(c0 + lane) * LLD + c0 + j] : (R)0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        R p = readlane_r(v[k], k);
        const bool bad = !(p > (R)0);  // wave-uniform
        first_bad = (bad && first_bad == 0) ? (c0 + k + 1) : first_bad;
        p = bad ? (R)1 : p;
        R rinv, dkk;
        rsqrt_sqrt(p, rinv, dkk);
        rk[k] = rinv;
        v[k] = (lane == k) ? dkk : v[k] * rinv;
#pragma unroll
        for (int j = k + 1; j < 16; ++j) {
          const R ljk = readlane_r(v[k], j);
          v[j] -= v[k] * ljk;
        }
      }
      if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) S[(c0 + lane) * LLD + c0 + j] = (j <= lane) ? v[j] : (R)0;
      }
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) rdiag[c0 + k] = rk[k];
      }
    } else if (wave != 0 && kb > 0 && !(dbg & 4)) {
      const int pc0 = c0 - 16;                 // previous panel
      const int pnt = 8 - kb;                  // its trailing matrix had pnt x pnt tiles; column 0 is done
      const int ndef = pnt * (pnt - 1) / 2;
      for (int t = wave - 1; t < ndef; t += 3) {
        int ta, tb;
        tri_decode(t, ta, tb);
        trailing_tile(c0 + 16 * (ta + 1), c0 + 16 * (tb + 1), pc0);
      }
    }
    __syncthreads();
    // ---- (2) rows below the diagonal block: X * Ld^T = P, one row per lane
    const int m = LEAF - c0 - 16;
    if (tid < m && !(dbg & 2)) {
      const int row = c0 + 16 + tid;
      R x[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) x[j] = S[row * LLD + c0 + j];
      // column-oriented substitution: once x[k] is final the later entries take their updates
      // independently of each other
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        x[k] *= rdiag[c0 + k];
#pragma unroll
        for (int j = k + 1; j < 16; ++j) x[j] -= x[k] * S[(c0 + j) * LLD + c0 + k];
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) S[row * LLD + c0 + j] = x[j];
    }
    __syncthreads();
    // ---- (3) rank-16 update of the FIRST tile column of the trailing matrix (the next diagonal
    //      block and the next panel); the other tiles wait for the next iteration's phase (1)
    const int nt = 7 - kb;
    for (int t = wave; t < ((dbg & 4) ? 0 : nt); t += 4) trailing_tile(c0 + 16 + 16 * t, c0 + 16, c0);
    __syncthreads();
  }
  if (wave == 0 && lane == 0 && first_bad != 0) atomicCAS(info, 0, info_base + first_bad);

  for (int e = tid; e < LEAF * CPR; e += LEAF_THREADS) {
    const int i = e / CPR, j = (e % CPR) * EPC;
    V v;
#pragma unroll
    for (int q = 0; q < EPC; ++q) v[q] = (j + q <= i) ? S[i * LLD + j + q] : (R)0;
    *reinterpret_cast<V*>(L + (int64_t)i * ldl + j) = v;
  }
  __syncthreads();

  // ======================= in-place inverse of the lower factor =======================
  // (I1) the eight 16 x 16 diagonal blocks: forward substitution on the identity, one column
  //      per lane (lanes 0-15: block 2*wave, lanes 16-31: block 2*wave+1)
  if (dbg & 32) return;
  if (lane < 32 && !(dbg & 8)) {
    const int b0 = 16 * (2 * wave + (lane >> 4));
    const int j = lane & 15;
    R x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      R acc = (i == j) ? (R)1 : (R)0;
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
    Acc acc[4];
    // pass a: tmp = L21 * X11   (X11 lower: k-blocks >= tb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = wave + 4 * q;
      acc[q] = acc_zero<R>();
      if (t < total) {
        const int mg = t / tiles_per_merge, tt = t % tiles_per_merge;
        const int ta = tt / tps, tb = tt % tps;
        const int o = mg * 2 * s;
        const int i0 = o + s + 16 * ta, j0 = o + 16 * tb;
        for (int kblk = tb; kblk < tps; ++kblk) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int k = o + 16 * kblk + 4 * kk + fq;
            const R a = S[(i0 + fr) * LLD + k];
            const R b = S[k * LLD + j0 + fr];
            acc[q] = Real<R>::mfma(a, b, acc[q]);
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
        for (int r = 0; r < 4; ++r) S[(i0 + Real<R>::crow(lane, r)) * LLD + j0 + fr] = acc[q][r];
      }
    }
    __syncthreads();
    // pass b: X21 = -X22 * tmp   (X22 lower: k-blocks <= ta)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = wave + 4 * q;
      acc[q] = acc_zero<R>();
      if (t < total) {
        const int mg = t / tiles_per_merge, tt = t % tiles_per_merge;
        const int ta = tt / tps, tb = tt % tps;
        const int o = mg * 2 * s;
        const int i0 = o + s + 16 * ta, j0 = o + 16 * tb;
        for (int kblk = 0; kblk <= ta; ++kblk) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int k = o + s + 16 * kblk + 4 * kk + fq;
            const R a = -S[(i0 + fr) * LLD + k];
            const R b = S[k * LLD + j0 + fr];
            acc[q] = Real<R>::mfma(a, b, acc[q]);
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
        for (int r = 0; r < 4; ++r) S[(i0 + Real<R>::crow(lane, r)) * LLD + j0 + fr] = acc[q][r];
      }
    }
    __syncthreads();
  }

  for (int e = tid; e < LEAF * CPR; e += LEAF_THREADS) {
    const int i = e / CPR, j = (e % CPR) * EPC;
    V v;
#pragma unroll
    for (int q = 0; q < EPC; ++q) v[q] = (j + q <= i) ? S[i * LLD + j + q] : (R)0;
    *reinterpret_cast<V*>(Linv + (int64_t)i * ldi + j) = v;
  }
}

template <typename R>
int launch_chol_leaf(const R* A, int64_t lda, R* L, int64_t ldl, R* Linv, int64_t ldi, int* info, int info_base,
                     hipStream_t s) {
  // the dynamic-LDS limit is a per-device function attribute: set it once per device (and per
  // template instance -- this static lives in launch_chol_leaf<R>)
  static std::atomic<bool> attr_set[64];
  constexpr size_t lds = sizeof(R) * (LEAF * LLD + LEAF);
  int device = 0;
  GP_HIP(hipGetDevice(&device));
  if (device < 0 || device >= 64 || !attr_set[device].load(std::memory_order_acquire)) {
    GP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_leaf_kernel<R>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (device >= 0 && device < 64) attr_set[device].store(true, std::memory_order_release);
  }
#ifdef GPFIT_DEV
  static int dbg = -1;  // phase-ablation switch for scripts/scratch/dev_leaf.py (timing only; wrong results)
  if (dbg < 0) dbg = getenv("GPFIT_LEAF_DBG") ? atoi(getenv("GPFIT_LEAF_DBG")) : 0;
#else
  constexpr int dbg = 0;
#endif
  hipLaunchKernelGGL(chol_leaf_kernel<R>, dim3(1), dim3(LEAF_THREADS), lds, s, A, lda, L, ldl, Linv, ldi, info,
                     info_base, dbg);
  GP_HIP(hipGetLastError());
  return 0;
}

template int launch_chol_leaf<double>(const double*, int64_t, double*, int64_t, double*, int64_t, int*, int,
                                      hipStream_t);
template int launch_chol_leaf<float>(const float*, int64_t, float*, int64_t, float*, int64_t, int*, int, hipStream_t);

}  // namespace gpfit
