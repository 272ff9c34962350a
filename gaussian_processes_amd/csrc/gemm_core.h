// fp64 MFMA GEMM main loop for gfx950 (CDNA4), shared by the plain GEMM and by the fused
// arc-cosine Gram kernel.  Internal header.
//
// Block = 256 threads = 4 waves (2 x 2), block tile 128 x 128, K step 16 through LDS,
// wave tile 64 x 64 = 4 x 4 tiles of v_mfma_f64_16x16x4_f64 (128 accumulator VGPRs).
// Operands are staged global -> registers -> LDS with a one-tile register prefetch and two
// LDS buffers (one barrier per K step).  Both operands sit in LDS "k-major"
// ([16][128] doubles, the M/N index contiguous) so that an MFMA fragment read is one
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
constexpr int LDS_TILE = KTILE * TILE;  // doubles per operand per buffer

__device__ __forceinline__ int lds_swz(int k) { return (((k >> 1) & 7) << 1) | ((k & 1) << 4); }

// Global -> registers: 4 x 16-byte chunks per thread for one 128 x 16 operand tile.
//   KMAJOR  : source element (x,k) at P[k*ld + x]  (x contiguous)  -> chunk = (k, x pair)
//   !KMAJOR : source element (x,k) at P[x*ld + k]  (k contiguous)  -> chunk = (x, k pair)
template <bool KMAJOR, bool EDGE>
__device__ __forceinline__ void tile_gload(v2d (&r)[4], const double* __restrict__ P, int64_t ld,
                                           int x0, int k0, int X, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + GEMM_THREADS * i;
    if (KMAJOR) {
      const int k = c >> 6, x = x0 + 2 * (c & 63);
      if (!EDGE || x < X) r[i] = *reinterpret_cast<const v2d*>(P + (int64_t)(k0 + k) * ld + x);
      else r[i] = v2d{0.0, 0.0};
    } else {
      const int x = x0 + (c >> 3), k = k0 + 2 * (c & 7);
      if (!EDGE || x < X) r[i] = *reinterpret_cast<const v2d*>(P + (int64_t)x * ld + k);
      else r[i] = v2d{0.0, 0.0};
    }
  }
}

// Registers -> LDS (swizzled k-major image).
template <bool KMAJOR>
__device__ __forceinline__ void tile_sstore(const v2d (&r)[4], double* __restrict__ S, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + GEMM_THREADS * i;
    if (KMAJOR) {
      const int k = c >> 6, x = 2 * (c & 63);
      *reinterpret_cast<v2d*>(S + k * TILE + (x ^ lds_swz(k))) = r[i];
    } else {
      const int x = c >> 3, k = 2 * (c & 7);
      S[k * TILE + (x ^ lds_swz(k))] = r[i].x;
      S[(k + 1) * TILE + (x ^ lds_swz(k + 1))] = r[i].y;
    }
  }
}

// acc[mi][ni] += op(A)[row0.., k] * op(B)[k, col0..] over k in [kbeg, kend) (multiples of 16).
// smem: 4 * LDS_TILE doubles (64 KiB): [buf][A|B][16][128].
template <bool A_KMAJOR, bool B_KMAJOR, bool EDGE>
__device__ __forceinline__ void gemm_mainloop(const double* __restrict__ A, int64_t lda,
                                              const double* __restrict__ B, int64_t ldb, int M, int N,
                                              int row0, int col0, int kbeg, int kend, double* smem,
                                              v4d (&acc)[4][4]) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;
  v2d ra[4], rb[4];

  if (kbeg >= kend) return;
  tile_gload<A_KMAJOR, EDGE>(ra, A, lda, row0, kbeg, M, tid);
  tile_gload<B_KMAJOR, EDGE>(rb, B, ldb, col0, kbeg, N, tid);
  tile_sstore<A_KMAJOR>(ra, smem, tid);
  tile_sstore<B_KMAJOR>(rb, smem + LDS_TILE, tid);
  __syncthreads();

  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += KTILE) {
    const bool more = (k0 + KTILE) < kend;
    if (more) {
      tile_gload<A_KMAJOR, EDGE>(ra, A, lda, row0, k0 + KTILE, M, tid);
      tile_gload<B_KMAJOR, EDGE>(rb, B, ldb, col0, k0 + KTILE, N, tid);
    }
    const double* As = smem + buf * 2 * LDS_TILE;
    const double* Bs = As + LDS_TILE;
#pragma unroll
    for (int kk = 0; kk < KTILE / 4; ++kk) {
      const int krow = kk * 4 + fk;
      const int sw = lds_swz(krow);
      double a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = As[krow * TILE + ((wm * 64 + i * 16 + fr) ^ sw)];
        b[i] = Bs[krow * TILE + ((wn * 64 + i * 16 + fr) ^ sw)];
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
    if (more) {
      double* Sn = smem + (buf ^ 1) * 2 * LDS_TILE;
      tile_sstore<A_KMAJOR>(ra, Sn, tid);
      tile_sstore<B_KMAJOR>(rb, Sn + LDS_TILE, tid);
    }
    __syncthreads();
    buf ^= 1;
  }
}

// Accumulator element (mi, ni, r) of this lane sits at
//   row = row0 + wm*64 + mi*16 + (lane>>4) + 4*r ,  col = col0 + wn*64 + ni*16 + (lane&15)
// (v_mfma_f64_16x16x4_f64 C/D layout: col = lane&15, row = (lane>>4) + 4*reg).
template <typename F>
__device__ __forceinline__ void for_each_acc(const v4d (&acc)[4][4], int row0, int col0, F&& f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + wm * 64 + mi * 16 + (lane >> 4) + 4 * r;
        const int col = col0 + wn * 64 + ni * 16 + (lane & 15);
        f(row, col, acc[mi][ni][r]);
      }
}

// Lower-triangular tile enumeration: t -> (ti, tj), tj <= ti.
__device__ __forceinline__ void tri_tile(int t, int& ti, int& tj) {
  int i = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((i + 1) * (i + 2) / 2 <= t) ++i;
  while (i * (i + 1) / 2 > t) --i;
  ti = i;
  tj = t - i * (i + 1) / 2;
}

}  // namespace gpfit
