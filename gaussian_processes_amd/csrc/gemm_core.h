// fp64 MFMA GEMM main loop for gfx950 (CDNA4), shared by the plain GEMM and by the fused
// arc-cosine Gram kernel.  Internal header.
//
// Block = 256 threads = 4 waves (2 x 2), block tile T x T (T = 128 for large problems; 64 / 32
// so that small panels of the recursive Cholesky still spread over many CUs), K step 16
// through LDS, wave tile T/2 x T/2 of v_mfma_f64_16x16x4_f64 tiles (4 x 4 = 128 accumulator
// VGPRs at T = 128).
// Operands are staged global -> registers -> LDS with a one-tile register prefetch and two
// LDS buffers (one barrier per K step).  Both operands sit in LDS "k-major"
// ([16][T] doubles, the M/N index contiguous) so that an MFMA fragment read is one
// ds_read_b64 per lane over 16 consecutive doubles per k row; the column index is
// XOR-swizzled with the k row so that (a) the two k rows a 32-lane half reads fall in
// different bank halves and (b) the 8 k-pairs x 2 rows a 16-lane group writes when a
// k-contiguous source is transposed on the way in fall on 32 distinct banks.
#pragma once
#include "common.h"

namespace gpfit {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int GEMM_THREADS = 256;

__device__ __forceinline__ int lds_swz(int k) { return (((k >> 1) & 7) << 1) | ((k & 1) << 4); }

// Global -> registers: T/32 x 16-byte chunks per thread for one T x 16 operand tile.
//   KMAJOR  : source element (x,k) at P[k*ld + x]  (x contiguous)  -> chunk = (k, x pair)
//   !KMAJOR : source element (x,k) at P[x*ld + k]  (k contiguous)  -> chunk = (x, k pair)
template <bool KMAJOR, bool EDGE, int T>
__device__ __forceinline__ void tile_gload(v2d (&r)[T / 32], const double* __restrict__ P, int64_t ld,
                                           int x0, int k0, int X, int tid) {
#pragma unroll
  for (int i = 0; i < T / 32; ++i) {
    const int c = tid + GEMM_THREADS * i;
    if (KMAJOR) {
      const int k = c / (T / 2), x = x0 + 2 * (c % (T / 2));
      if (!EDGE || x < X) r[i] = *reinterpret_cast<const v2d*>(P + (int64_t)(k0 + k) * ld + x);
      else r[i] = v2d{0.0, 0.0};
    } else {
      const int x = x0 + (c >> 3), k = k0 + 2 * (c & 7);
      if (!EDGE || x < X) r[i] = *reinterpret_cast<const v2d*>(P + (int64_t)x * ld + k);
      else r[i] = v2d{0.0, 0.0};
    }
  }
}

// Registers -> LDS (swizzled k-major image [16][T]).
template <bool KMAJOR, int T>
__device__ __forceinline__ void tile_sstore(const v2d (&r)[T / 32], double* __restrict__ S, int tid) {
#pragma unroll
  for (int i = 0; i < T / 32; ++i) {
    const int c = tid + GEMM_THREADS * i;
    if (KMAJOR) {
      const int k = c / (T / 2), x = 2 * (c % (T / 2));
      *reinterpret_cast<v2d*>(S + k * T + (x ^ lds_swz(k))) = r[i];
    } else {
      const int x = c >> 3, k = 2 * (c & 7);
      S[k * T + (x ^ lds_swz(k))] = r[i].x;
      S[(k + 1) * T + (x ^ lds_swz(k + 1))] = r[i].y;
    }
  }
}

// acc[mi][ni] += op(A)[row0.., k] * op(B)[k, col0..] over k in [kbeg, kend) (multiples of 16).
// Block tile T x T (T = 128, 64 or 32), 4 waves as 2 x 2, wave tile T/2 x T/2.
// smem: 4 * 16 * T doubles: [buf][A|B][16][T].
template <bool A_KMAJOR, bool B_KMAJOR, bool EDGE, int T>
__device__ __forceinline__ void gemm_mainloop(const double* __restrict__ A, int64_t lda,
                                              const double* __restrict__ B, int64_t ldb, int M, int N,
                                              int row0, int col0, int kbeg, int kend, double* smem,
                                              v4d (&acc)[T / 32][T / 32]) {
  constexpr int MI = T / 32, WT = T / 2, LT = KTILE * T;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;
  v2d ra[MI], rb[MI];

  if (kbeg >= kend) return;
  tile_gload<A_KMAJOR, EDGE, T>(ra, A, lda, row0, kbeg, M, tid);
  tile_gload<B_KMAJOR, EDGE, T>(rb, B, ldb, col0, kbeg, N, tid);
  tile_sstore<A_KMAJOR, T>(ra, smem, tid);
  tile_sstore<B_KMAJOR, T>(rb, smem + LT, tid);
  __syncthreads();

  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += KTILE) {
    const bool more = (k0 + KTILE) < kend;
    if (more) {
      tile_gload<A_KMAJOR, EDGE, T>(ra, A, lda, row0, k0 + KTILE, M, tid);
      tile_gload<B_KMAJOR, EDGE, T>(rb, B, ldb, col0, k0 + KTILE, N, tid);
    }
    const double* As = smem + buf * 2 * LT;
    const double* Bs = As + LT;
#pragma unroll
    for (int kk = 0; kk < KTILE / 4; ++kk) {
      const int krow = kk * 4 + fk;
      const int sw = lds_swz(krow);
      double a[MI], b[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        a[i] = As[krow * T + ((wm * WT + i * 16 + fr) ^ sw)];
        b[i] = Bs[krow * T + ((wn * WT + i * 16 + fr) ^ sw)];
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < MI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
    if (more) {
      double* Sn = smem + (buf ^ 1) * 2 * LT;
      tile_sstore<A_KMAJOR, T>(ra, Sn, tid);
      tile_sstore<B_KMAJOR, T>(rb, Sn + LT, tid);
    }
    __syncthreads();
    buf ^= 1;
  }
}

// Accumulator element (mi, ni, r) of this lane sits at
//   row = row0 + wm*T/2 + mi*16 + (lane>>4) + 4*r ,  col = col0 + wn*T/2 + ni*16 + (lane&15)
// (v_mfma_f64_16x16x4_f64 C/D layout: col = lane&15, row = (lane>>4) + 4*reg).
template <int T, typename F>
__device__ __forceinline__ void for_each_acc(const v4d (&acc)[T / 32][T / 32], int row0, int col0, F&& f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int mi = 0; mi < T / 32; ++mi)
#pragma unroll
    for (int ni = 0; ni < T / 32; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + wm * (T / 2) + mi * 16 + (lane >> 4) + 4 * r;
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
  tri_tile(t / (r * r), bi, rem);  // rem unused: just the block row of the r*r-sized group
  // groups of r*r tiles are ordered by block pairs; recompute exactly from the block-row start
  const int start = r * r * (bi * (bi + 1) / 2);
  const int v = t - start;
  const int w = (bi + 1) * r;
  ti = bi * r + v / w;
  tj = v % w;
}
__host__ __device__ inline int lower_tile_count(int nblocks, int r) { return r * r * (nblocks * (nblocks + 1) / 2); }

}  // namespace gpfit
