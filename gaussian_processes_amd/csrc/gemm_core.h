// MFMA GEMM main loop for gfx950 (CDNA4), shared by the plain GEMM, the stream-K GEMM and the
// fused arc-cosine Gram kernel; fp64 (v_mfma_f64_16x16x4_f64) and fp32 (v_mfma_f32_16x16x4_f32)
// through the Real<> traits.  Internal header.
//
// Block = 256 threads = 4 waves (2 x 2), block tile T x T (T = 128 for large problems; 64 / 32
// so that small panels of the recursive Cholesky still spread over many CUs), K step of
// 16 (fp64) / 32 (fp32) staged through LDS, wave tile T/2 x T/2 of 16 x 16 x 4 MFMA tiles
// (4 x 4 accumulators at T = 128).
// Operands are staged global -> registers -> LDS with a one-tile register prefetch and two
// LDS buffers (one barrier per K step).  Both operands sit in LDS "k-major" ([KT][T], the M/N
// index contiguous) so that an MFMA fragment read is one ds_read per lane over 16 consecutive
// elements per k row; the column index is XOR-swizzled with the k row so that (a) the two k rows
// a 32-lane half reads fall in different bank halves and (b) the stores that transpose a
// k-contiguous source on the way in spread over the banks.
#pragma once
#include "common.h"

namespace gpfit {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int GEMM_THREADS = 256;

// Per-type pieces: accumulator / 16-byte vector types, elements per 16-byte chunk, K step,
// the MFMA itself, the C/D register -> row map and the LDS swizzle.
template <typename R> struct Real;
template <> struct Real<double> {
  using acc_t = v4d;
  using vec_t = v2d;
  static constexpr int EPC = 2, KT = 16;
  static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
  static __device__ __forceinline__ int swz(int k) { return (((k >> 1) & 7) << 1) | ((k & 1) << 4); }
};
template <> struct Real<float> {
  using acc_t = v4f;
  using vec_t = v4f;
  static constexpr int EPC = 4, KT = 32;
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 * (lane >> 4) + reg
  static __device__ __forceinline__ int crow(int lane, int r) { return 4 * (lane >> 4) + r; }
  // multiples of 4 keep every 16-byte chunk whole; bit 4 separates the two k rows of a half-wave
  static __device__ __forceinline__ int swz(int k) { return (((k >> 1) & 3) << 2) | ((k & 1) << 4); }
};

template <typename R> __device__ __forceinline__ typename Real<R>::acc_t acc_zero() {
  typename Real<R>::acc_t z = {0, 0, 0, 0};
  return z;
}

// Global -> registers: T/32 x 16-byte chunks per thread for one T x KT operand tile.
//   KMAJOR  : source element (x,k) at P[k*ld + x]  (x contiguous)  -> chunk = (k, EPC x's)
//   !KMAJOR : source element (x,k) at P[x*ld + k]  (k contiguous)  -> chunk = (x, EPC k's)
template <typename R, bool KMAJOR, bool EDGE, int T>
__device__ __forceinline__ void tile_gload(typename Real<R>::vec_t (&r)[T / 32], const R* __restrict__ P, int64_t ld,
                                           int x0, int k0, int X, int tid) {
  using V = typename Real<R>::vec_t;
  constexpr int EPC = Real<R>::EPC;
#pragma unroll
  for (int i = 0; i < T / 32; ++i) {
    const int c = tid + GEMM_THREADS * i;
    if (KMAJOR) {
      const int k = c / (T / EPC), x = x0 + EPC * (c % (T / EPC));
      if (!EDGE || x < X) r[i] = *reinterpret_cast<const V*>(P + (int64_t)(k0 + k) * ld + x);
      else r[i] = V{};
    } else {
      const int x = x0 + (c >> 3), k = k0 + EPC * (c & 7);
      if (!EDGE || x < X) r[i] = *reinterpret_cast<const V*>(P + (int64_t)x * ld + k);
      else r[i] = V{};
    }
  }
}

// Registers -> LDS (swizzled k-major image [KT][T]).
template <typename R, bool KMAJOR, int T>
__device__ __forceinline__ void tile_sstore(const typename Real<R>::vec_t (&r)[T / 32], R* __restrict__ S, int tid) {
  using V = typename Real<R>::vec_t;
  constexpr int EPC = Real<R>::EPC;
#pragma unroll
  for (int i = 0; i < T / 32; ++i) {
    const int c = tid + GEMM_THREADS * i;
    if (KMAJOR) {
      const int k = c / (T / EPC), x = EPC * (c % (T / EPC));
      *reinterpret_cast<V*>(S + k * T + (x ^ Real<R>::swz(k))) = r[i];
    } else {
      const int x = c >> 3, k = EPC * (c & 7);
#pragma unroll
      for (int e = 0; e < EPC; ++e) S[(k + e) * T + (x ^ Real<R>::swz(k + e))] = r[i][e];
    }
  }
}

// acc[mi][ni] += op(A)[row0.., k] * op(B)[k, col0..] over k in [kbeg, kend) (multiples of KT).
// Block tile T x T (T = 128, 64 or 32), 4 waves as 2 x 2, wave tile T/2 x T/2.
// smem: 4 * KT * T elements: [buf][A|B][KT][T].
template <typename R, bool A_KMAJOR, bool B_KMAJOR, bool EDGE, int T>
__device__ __forceinline__ void gemm_mainloop(const R* __restrict__ A, int64_t lda, const R* __restrict__ B,
                                              int64_t ldb, int M, int N, int row0, int col0, int kbeg, int kend,
                                              R* smem, typename Real<R>::acc_t (&acc)[T / 32][T / 32]) {
  using V = typename Real<R>::vec_t;
  constexpr int MI = T / 32, WT = T / 2, KT = Real<R>::KT, LT = KT * T;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;
  V ra[MI], rb[MI];

  if (kbeg >= kend) return;
  tile_gload<R, A_KMAJOR, EDGE, T>(ra, A, lda, row0, kbeg, M, tid);
  tile_gload<R, B_KMAJOR, EDGE, T>(rb, B, ldb, col0, kbeg, N, tid);
  tile_sstore<R, A_KMAJOR, T>(ra, smem, tid);
  tile_sstore<R, B_KMAJOR, T>(rb, smem + LT, tid);
  __syncthreads();

  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += KT) {
    const bool more = (k0 + KT) < kend;
    if (more) {
      tile_gload<R, A_KMAJOR, EDGE, T>(ra, A, lda, row0, k0 + KT, M, tid);
      tile_gload<R, B_KMAJOR, EDGE, T>(rb, B, ldb, col0, k0 + KT, N, tid);
    }
    const R* As = smem + buf * 2 * LT;
    const R* Bs = As + LT;
#pragma unroll
    for (int kk = 0; kk < KT / 4; ++kk) {
      const int krow = kk * 4 + fk;
      const int sw = Real<R>::swz(krow);
      R a[MI], b[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        a[i] = As[krow * T + ((wm * WT + i * 16 + fr) ^ sw)];
        b[i] = Bs[krow * T + ((wn * WT + i * 16 + fr) ^ sw)];
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < MI; ++ni) acc[mi][ni] = Real<R>::mfma(a[mi], b[ni], acc[mi][ni]);
    }
    if (more) {
      R* Sn = smem + (buf ^ 1) * 2 * LT;
      tile_sstore<R, A_KMAJOR, T>(ra, Sn, tid);
      tile_sstore<R, B_KMAJOR, T>(rb, Sn + LT, tid);
    }
    __syncthreads();
    buf ^= 1;
  }
}

// Accumulator element (mi, ni, r) of this lane sits at
//   row = row0 + wm*T/2 + mi*16 + crow(lane, r) ,  col = col0 + wn*T/2 + ni*16 + (lane & 15)
template <typename R, int T, typename F>
__device__ __forceinline__ void for_each_acc(const typename Real<R>::acc_t (&acc)[T / 32][T / 32], int row0,
                                             int col0, F&& f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int mi = 0; mi < T / 32; ++mi)
#pragma unroll
    for (int ni = 0; ni < T / 32; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + wm * (T / 2) + mi * 16 + Real<R>::crow(lane, r);
        const int col = col0 + wn * (T / 2) + ni * 16 + (lane & 15);
        f(row, col, acc[mi][ni][r]);
      }
}

// Lower-triangular tile enumeration: t -> (ti, tj), tj <= ti.
__device__ __host__ inline void tri_tile(int t, int& ti, int& tj) {
  int i = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((i + 1) * (i + 2) / 2 <= t) ++i;
  while (i * (i + 1) / 2 > t) --i;
  ti = i;
  tj = t - i * (i + 1) / 2;
}

// "Lower" output region = the 128 x 128 blocks on/below the block diagonal, whatever the block
// tile T (so that the bytes written do not depend on the tile the launcher picked): with
// r = 128/T sub-tiles per block side, block row bi holds r tile rows of (bi+1)*r tiles.
__device__ __host__ inline void lower_tile(int t, int r, int& ti, int& tj) {
  int bi, rem;
  tri_tile(t / (r * r), bi, rem);  // block row of the r*r-sized group t falls in
  const int start = r * r * (bi * (bi + 1) / 2);
  const int v = t - start;
  const int w = (bi + 1) * r;
  ti = bi * r + v / w;
  tj = v % w;
}
__host__ __device__ inline int lower_tile_count(int nblocks, int r) { return r * r * (nblocks * (nblocks + 1) / 2); }

}  // namespace gpfit
