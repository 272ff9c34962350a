// Fused arc-cosine Gram kernel: the N x N x d MFMA GEMM  G = (X C) X^T + s0^2  with the
// whole element-wise chain of the reference's acosker (utils.py:978-990) applied to the
// accumulators in registers, so K~ is written to HBM exactly once and the ~100 separate
// N x N passes of the reference never exist.
//   c = clip(G / (q_i q_j + 1e-7), -1, 1)                 utils.py:984
//   delta = acos(c)                                       utils.py:986
//   J = (sqrt(1 - c^2) + pi c - delta c) / pi             utils.py:988   (pi = float32 pi)
//   K = q_i q_j J                                         utils.py:990
// For the square case only the tiles on/below the diagonal are computed (K~ is symmetric;
// the reference symmetrises it explicitly, utils.py:1024-1025).  Storage convention for every
// symmetric N x N matrix of the library: the lower triangle (i >= j) is canonical; the strict
// upper part of a diagonal tile is written but never read, tiles above the diagonal are not
// written at all (gpfit_symmetrize mirrors when a caller wants the full matrix).
#include "gemm_core.h"

namespace gpfit {

__global__ __launch_bounds__(GEMM_THREADS, 2) void gram_acos_kernel(GramArgs p, int tiles_n) {
  __shared__ __attribute__((aligned(16))) double smem[4 * KTILE * TILE];
  int ti, tj;
  if (p.lower) {
    tri_tile(blockIdx.x, ti, tj);
  } else {
    ti = blockIdx.x / tiles_n;
    tj = blockIdx.x % tiles_n;
  }
  const int row0 = ti * TILE, col0 = tj * TILE;
  v4d acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = v4d{0.0, 0.0, 0.0, 0.0};

  // operands are zero-padded to whole tiles: no edge predication on the loads
  gemm_mainloop<true, true, false, TILE>(p.XCt, p.ld1, p.Xt, p.ld2, p.np1, p.np2, row0, col0, 0, p.Kd, smem, acc);

  const int nv1 = p.nv1, nv2 = p.nv2;
  const bool pad_id = p.pad_identity != 0;
  const int64_t ldk = p.ldk;
  const int64_t ldcos = p.ldcos ? p.ldcos : p.ldk;
  const double s0sq = p.s0sq;
  const double* __restrict__ q1 = p.q1;
  const double* __restrict__ q2 = p.q2;
  double* __restrict__ Ko = p.Kout;
  double* __restrict__ Co = p.Cos;

  for_each_acc<TILE>(acc, row0, col0, [&](int row, int col, double g) {
    const int64_t o = (int64_t)row * ldk + col;
    const int64_t oc = (int64_t)row * ldcos + col;
    if (row >= nv1 || col >= nv2) {
      // padding: identity on the diagonal so the padded matrix factorises as [L 0; 0 I]
      if (pad_id) {
        Ko[o] = (row == col) ? 1.0 : 0.0;
        if (Co) Co[oc] = 0.0;
      }
      return;
    }
    const double qq = q1[row] * q2[col];
    double c = (g + s0sq) / (qq + 1e-7);
    c = fmin(1.0, fmax(-1.0, c));
    const double delta = acos(c);
    const double J = (sqrt(1.0 - c * c) + PI32 * c - delta * c) / PI32;
    Ko[o] = qq * J;
    if (Co) Co[oc] = c;
  });
}

int launch_gram(const GramArgs& a, hipStream_t s) {
  if (a.nv1 <= 0 || a.nv2 <= 0) return 0;
  if (a.Kd % KTILE != 0 || (a.np1 % TILE) || (a.np2 % TILE) || (a.ld1 & 1) || (a.ld2 & 1)) {
    set_error("launch_gram: Kd must be a multiple of 16, np1/np2 multiples of 128, ld1/ld2 even");
    return -3;
  }
  if (a.lower && a.np1 != a.np2) {
    set_error("launch_gram: lower needs a square problem");
    return -3;
  }
  // only tiles that contain valid (or identity-padded) output are launched
  const int tm = (a.pad_identity ? a.np1 : (int)round_up(a.nv1, TILE)) / TILE;
  const int tn = (a.pad_identity ? a.np2 : (int)round_up(a.nv2, TILE)) / TILE;
  const int tiles = a.lower ? tm * (tm + 1) / 2 : tm * tn;
  hipLaunchKernelGGL(gram_acos_kernel, dim3(tiles), dim3(GEMM_THREADS), 0, s, a, tn);
  GP_HIP(hipGetLastError());
  return 0;
}

}  // namespace gpfit
