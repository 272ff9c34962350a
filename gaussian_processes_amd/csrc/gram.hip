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
// Instantiated for fp64 (the reference's precision) and fp32 (theta-grid configuration).
#include "gemm_core.h"

namespace gpfit {

template <typename R>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gram_acos_kernel(GramArgsT<R> p, int tiles_n) {
  __shared__ __attribute__((aligned(16))) R smem[4 * Real<R>::KT * TILE];
  int ti, tj;
  if (p.lower) {
    tri_tile(blockIdx.x, ti, tj);
  } else {
    ti = blockIdx.x / tiles_n;
    tj = blockIdx.x % tiles_n;
  }
  const int row0 = ti * TILE, col0 = tj * TILE;
  typename Real<R>::acc_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = acc_zero<R>();

  // operands are zero-padded to whole tiles: no edge predication on the loads
  gemm_mainloop<R, true, true, false, TILE>(p.XCt, p.ld1, p.Xt, p.ld2, p.np1, p.np2, row0, col0, 0, p.Kd, smem, acc);

  const int nv1 = p.nv1, nv2 = p.nv2;
  const bool pad_id = p.pad_identity != 0;
  const int64_t ldk = p.ldk;
  const int64_t ldcos = p.ldcos ? p.ldcos : p.ldk;
  const R s0sq = (R)p.s0sq;
  const R pi32 = (R)PI32;
  const R* __restrict__ q1 = p.q1;
  const R* __restrict__ q2 = p.q2;
  R* __restrict__ Ko = p.Kout;
  R* __restrict__ Co = p.Cos;
  const bool mirror = p.lower && p.mirror;

  for_each_acc<R, TILE>(acc, row0, col0, [&](int row, int col, R g) {
    const int64_t o = (int64_t)row * ldk + col;
    const int64_t oc = (int64_t)row * ldcos + col;
    if (mirror && row < col) return;   // the strict upper part of a diagonal tile comes from its mirrored lower part
    if (row >= nv1 || col >= nv2) {
      // padding: identity on the diagonal so the padded matrix factorises as [L 0; 0 I]
      if (pad_id) {
        Ko[o] = (row == col) ? (R)1 : (R)0;
        if (mirror && row > col) Ko[(int64_t)col * ldk + row] = (R)0;
        if (Co) Co[oc] = (R)0;
      }
      return;
    }
    const R qq = q1[row] * q2[col];
    R c = (g + s0sq) / (qq + (R)1e-7);
    c = fmin((R)1, fmax((R)-1, c));
    const R delta = acos(c);
    const R J = (sqrt((R)1 - c * c) + pi32 * c - delta * c) / pi32;
    Ko[o] = qq * J;
    if (mirror && row > col) Ko[(int64_t)col * ldk + row] = qq * J;
    if (Co) Co[oc] = c;
  });
}

template <typename R>
int launch_gram(const GramArgsT<R>& a, hipStream_t s) {
  if (a.nv1 <= 0 || a.nv2 <= 0) return 0;
  constexpr int EPC = 16 / (int)sizeof(R);
  if (a.Kd % ktile_of<R>() != 0 || (a.np1 % TILE) || (a.np2 % TILE) || (a.ld1 % EPC) || (a.ld2 % EPC)) {
    set_error("launch_gram: Kd must be a multiple of the K step, np1/np2 multiples of 128, ld1/ld2 of 16 bytes");
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
  hipLaunchKernelGGL(gram_acos_kernel<R>, dim3(tiles), dim3(GEMM_THREADS), 0, s, a, tn);
  GP_HIP(hipGetLastError());
  return 0;
}

template int launch_gram<double>(const GramArgsT<double>&, hipStream_t);
template int launch_gram<float>(const GramArgsT<float>&, hipStream_t);

}  // namespace gpfit
