// Shared declarations for the MI355X (gfx950) GP-fit library.  Internal header.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

namespace gpfit {

// float32-rounded pi: the reference overwrites torch.pi before switching the default
// dtype to float64 (reference utils.py:25 vs :33), so every pi in its kernel is this value.
constexpr double PI32 = 3.1415927410125732;

constexpr int TILE = 128;  // GEMM block tile (M and N) and Cholesky leaf size
constexpr int TRMV_ROWS = 128;  // rows per block of the transposed triangular matrix-vector product (partials: np / TRMV_ROWS slices)
constexpr int KTILE = 16;  // fp64 GEMM K step staged through LDS (fp32: 32, see ktile_of)

void set_error(const std::string& msg);

#define GP_HIP(expr)                                                                     \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      gpfit::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));               \
      return -100;                                                                       \
    }                                                                                    \
  } while (0)

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// ---- MFMA GEMM (fp64, and fp32 for the theta-grid configuration) ----------------------
// C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] + beta * C        (row-major storage)
//   a_kmajor = 0 : A stored [M][K] (k contiguous)   element (m,k) at A[m*lda + k]
//   a_kmajor = 1 : A stored [K][M] (m contiguous)   element (m,k) at A[k*lda + m]
//   b_kmajor = 1 : B stored [K][N] (n contiguous)   element (k,n) at B[k*ldb + n]
//   b_kmajor = 0 : B stored [N][K] (k contiguous)   element (k,n) at B[n*ldb + k]
// Triangular structure is exploited per block tile:
//   out_lower   : only the 128-blocks on/below the block diagonal are computed / written (M == N)
//   a_tri/b_tri : 0 dense, 1 op() is lower triangular, 2 op() is upper triangular
//                 (restricts each tile's k range; the skipped part must hold zeros or is
//                  simply never read)
// K must be a multiple of the K step (16 for fp64, 32 for fp32); M, N arbitrary (edges are
// predicated); lda/ldb multiples of 16 bytes.
constexpr int GEMM_MAXB = 32;  // problems of one pointer-batched launch (GemmArgsT::nptr)

template <typename R>
struct GemmArgsT {
  const R* A;
  const R* B;
  R* C;
  int64_t lda, ldb, ldc;
  int M, N, K;
  double alpha, beta;
  int a_kmajor, b_kmajor;
  int out_lower;
  int a_tri, b_tri;
  int batch;                 // number of independent problems (grid.y)
  int64_t sA, sB, sC;        // batch strides (elements)
  int split_k;               // >1: partial products written to C + z*sC (beta ignored)
  int tile;                  // 0 = choose (128 / 64 / 32), else forced block tile
  int reverse;               // tile walk: bit 0 backwards, bit 1 column-major (dense output)
  int workspace;             // stream-K partial-tile workspace to use (0 main stream, 1 aux stream)
  void* sk_ws;               // caller-owned stream-K workspace (>= SK_WS_BYTES); nullptr: process-wide one
  int tile_limit;            // >0: launch only the first tile_limit tiles of the walk (stream-K head)
  int half_occ;              // 1: pad the launch with unused dynamic LDS so that only ONE workgroup of it fits on
                             // a CU: a long GEMM off the critical path then leaves half of every CU (79 KiB
                             // of LDS, 320 VGPRs) to the latency-bound kernels of the critical chain instead
                             // of holding every workgroup slot of the chip for its whole duration
  const int* sched;          // device tile table of an XCD-aware schedule (gemm_sched.hip), or nullptr
  int sched_blocks;          // its length = the grid size
  // Fused epilogues of the data-parallel launches (gemm.hip; a launch that takes the stream-K schedule cannot
  // honour them -- ask gemm_epilogue_ok() first).  Bit mask:
  //   1  mirror: square lower output, every stored element below the diagonal is also stored transposed
  //      (replaces a symmetrisation pass over the matrix)
  //   2  tile norms: sumsq[ti (ti + 1) / 2 + tj] = sum of squares of the stored values of 128-tile (ti, tj)
  //      (lower output, 128-tile launches only; replaces a pass over the matrix; on the stream-K schedule split
  //      tiles leave theirs per fix-up band behind the per-tile table: gemm_sumsq_entries)
  //   4  dual update: with D = aux (same leading dimension as C), C = alpha op(A) op(B) + D and then
  //      aux = C + D (beta is ignored; replaces a copy and an axpby pass)
  int epi;
  R* aux;
  double* sumsq;
  // Pointer batch: nptr > 0 independent problems of the same shape, problem b (= blockIdx.y) on Ap[b], Bp[b],
  // Cp[b] instead of A, B, C and the strides.  This is how the latency-bound levels of several Cholesky
  // recursions (the K~ and V chains of one unit, the chains of several units in flight) and the sibling blocks of
  // the two-sided product share their launches: the matrices live in separately allocated workspaces, so there is
  // no common stride.  Data-parallel launches only (no stream-K, no schedule table); fused epilogues take their
  // per-problem operands from auxp / sumsqp.
  int nptr;
  const R* Ap[GEMM_MAXB];
  const R* Bp[GEMM_MAXB];
  R* Cp[GEMM_MAXB];
  R* auxp[GEMM_MAXB];        // epi 4: problem b's aux
  double* sumsqp[GEMM_MAXB]; // epi 2: problem b's tile-norm table
};
using GemmArgs = GemmArgsT<double>;

// K step staged through LDS: 16 KiB per operand per stage at T = 128 for either type
constexpr size_t SK_WS_BYTES = (size_t)2 * 512 * TILE * TILE * sizeof(double);  // 2 partial tiles per resident workgroup

template <typename R> constexpr int ktile_of() { return 128 / (int)sizeof(R); }

template <typename R> int launch_gemm(const GemmArgsT<R>& a, hipStream_t s);
template <typename R> int gemm_pick_tile(const GemmArgsT<R>& a);   // block tile the launcher will use (128 / 64 / 32)
// stream-K schedule for large 128-tile launches (gemm_streamk.hip): 0 issued, 1 not applicable
template <typename R> int launch_gemm_streamk(const GemmArgsT<R>& a, hipStream_t s);
template <typename R> int launch_gemm_plain(const GemmArgsT<R>& a, hipStream_t s);  // data-parallel launch, no stream-K
// Two independent pointer-batched products as ONE launch (blockIdx.z picks the problem): small-tile products of the
// latency-bound levels of the recursion that depend on the same predecessor but differ in structure (lower output or
// not, operand layout, alpha / beta), so they cannot share a pointer batch.  gemm_pair_ok: both are plain pointer
// batches on full 32-tiles with a row-major op(A); the launch then runs exactly the tile bodies the two separate
// launches would have run (same bits).
template <typename R> bool gemm_pair_ok(const GemmArgsT<R>& a, const GemmArgsT<R>& b);
template <typename R> int launch_gemm_pair(const GemmArgsT<R>& a, const GemmArgsT<R>& b, hipStream_t s);
// XCD-aware data-parallel schedule (gemm_sched.hip): 0 issued, 1 not applicable.  Walk bit 3 asks for it.
template <typename R> int launch_gemm_xcd(const GemmArgsT<R>& a, hipStream_t s);
// would launch_gemm run these arguments on a data-parallel schedule (plain or XCD-aware), i.e. honour a.epi?
template <typename R> bool gemm_epilogue_ok(const GemmArgsT<R>& a);
template <typename R> bool gemm_streamk_applies(const GemmArgsT<R>& a);   // gemm_streamk.hip
template <typename R> bool gemm_streamk_carries(const GemmArgsT<R>& a);   // gemm_streamk.hip: it would also honour a.epi
// entries of a.sumsq an epi-2 launch of these arguments writes (0: the launch cannot carry the epilogue): nt =
// tiles of the lower output on the data-parallel schedules (one per tile), 33 nt on the stream-K schedule (one per
// tile + one per fix-up band); the squared Frobenius norm is the sum over all of them either way
template <typename R> int gemm_sumsq_entries(const GemmArgsT<R>& a);
template <typename R> bool gemm_xcd_applies(const GemmArgsT<R>& a);       // gemm_sched.hip

// Arc-cosine Gram matrix from the k-major, zero-padded operands XCt[Kd][ld1], Xt[Kd][ld2]:
//   G = XCt^T Xt + s0^2 ; c = clip(G/(q1 q2 + 1e-7)) ; K = q1 q2 J(c)
// np1/np2: padded extents (multiples of 128) that the loads may touch; nv1/nv2: valid
// extents that are stored.  `lower`: only tiles on/below the diagonal (square case).
// `pad_identity`: rows/cols >= nv get the identity (Kout must then hold np1 x np2).
template <typename R>
struct GramArgsT {
  const R* XCt;
  const R* Xt;
  const R* q1;   // [np1]
  const R* q2;   // [np2]
  R* Kout;       // [..][ldk]
  R* Cos;        // same shape as Kout, or nullptr
  int64_t ld1, ld2, ldk;
  int64_t ldcos;      // leading dimension of Cos (0: same as ldk)
  int np1, np2, nv1, nv2, Kd;
  double s0sq;
  int lower;
  int pad_identity;
  int mirror;    // lower only: every element below the diagonal is also stored transposed, so Kout holds the full
                 // symmetric matrix (for callers that multiply with it: no separate symmetrisation pass); Cos stays lower
};
using GramArgs = GramArgsT<double>;
template <typename R> int launch_gram(const GramArgsT<R>& a, hipStream_t s);

}  // namespace gpfit
