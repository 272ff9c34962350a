// Cholesky leaf, register-resident variant: factor one 128 x 128 diagonal block and invert the
// factor with the matrix held in the accumulator registers of four tile waves and only the current
// 16-column panel in LDS (48 KiB instead of the 133 KiB of chol_leaf.hip); a fifth wave runs the pivot
// chains one panel ahead.
//
// Why: the leaf sits on the critical path of both factorisation chains 2N/128 times per fit (64 launches of
// two workgroups at the headline; a third of a unit's time at N = 2048).
//
// Layout.  The block is cut into 8 x 8 tiles of 16 x 16; tile (i, j), i >= j, lives in the MFMA
// C/D layout (4 values per lane) in the registers of the wave that owns tile COLUMN j (rl_col_a / rl_col_b
// below: wave 0, which shares its SIMD with the pivot wave, owns columns 6 and 7 = 3 tiles, the others two
// columns of 11 tiles each).  Column ownership is what makes the in-place inverse
// possible: for the fp64 16x16x4 MFMA the C/D register r of a tile holds exactly the rows
// k_r(lane) = (lane >> 4) + 4 r that the B operand of sub-step r needs (fp32: 4 (lane >> 4) + r, the
// sum over k simply runs in that order), so a register tile X[k, j] is used directly as the B
// operand of  Y[i, j] += L[i, k] X[k, j]  -- no lane movement -- as long as the same wave holds
// both, i.e. as long as tiles of one column stay together.
//
// Right-looking over the eight 16-column panels kb:
//   (1) pivot wave: the 16 x 16 diagonal block of panel kb with a row per lane, pivots and multipliers moving
//       between lanes by DPP row broadcasts fused into the FMAs; the same sweep inverts the factor (Dinv), so the
//       panel solve below is an MFMA product.  The wave works one panel ahead: behind barrier B1(kb) it forms
//       its own copy of L[kb + 1, kb], applies the last Schur update to the diagonal tile of column kb + 1 (handed
//       over through an LDS mailbox) and runs the chain of panel kb + 1 beside the tile waves' update pass;
//   (2) tile waves: the rows below the diagonal block,  L[i, kb] = S[i, kb] Dinv^T,  two tiles per wave,
//       written to the LDS panel P and to global memory; barrier B2 (tile waves only, see b2_arrivals);
//   (3) every tile wave updates its own tiles from the panel (leaf_update_pass, specialised per wave):
//         columns j > kb (still Schur complement):  S[i, j] -= L[i, kb] L[j, kb]^T
//         columns j <= kb (already inverse):        X[kb, j] = -Dinv Y[kb, j]   (X[kb, kb] = Dinv)
//                                                   Y[i, j] += L[i, kb] X[kb, j]   for i > kb
//       so the registers of column j hold S[., j] until panel j has been factored and the rows of
//       L^-1 (finished rows X, running sums Y) afterwards: the inverse costs no extra storage and
//       is complete when the last panel is.
// Rounds: 48.7 us (LDS-resident, r01) -> 45 (registers, r02) -> 36.6 (pivot wave, r03) -> 28.8 us (r04: update pass
// specialised per wave with its operands requested ahead, 3 / 11 / 11 / 11 tiles, the first chain under the tile
// waves' prologue, B2 without the pivot wave); per panel now 1.6 k cycles before the chain, 4.4 k chain, 1.5 k
// write-back on the pivot wave against 2.3 k solve + 5.4 k update pass on the tile waves.
// info: LAPACK-style, as chol_leaf.hip (first non-positive pivot, offending pivot replaced by 1).
#include "gemm_core.h"
#include "kernels.h"

#include <cstdlib>

namespace gpfit {

// scripts/scratch/dev_leaf_time.hip defines GPFIT_LEAF_STAMPS and includes this file: shader-clock stamps of
// wave `owner` / wave 0 at the phase boundaries of every panel (never compiled into the library)
#ifdef GPFIT_LEAF_STAMPS
__device__ long long g_leaf_stamps[9 * 8];
__device__ long long g_leaf_wave_end[8 * 4];     // end of the update pass (3) per panel and tile wave
#ifdef GPFIT_LEAF_STAMPS_MIN   // only the arrivals at B1 (the stamps in front of a barrier cost nothing extra) and kernel begin / end
#define LEAF_STAMP(kb, ph) do { if (((ph) == 3 || (kb) == 8) && (threadIdx.x & 63) == 0) g_leaf_stamps[(kb) * 8 + (ph)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define LEAF_STAMP(kb, ph) do { if ((threadIdx.x & 63) == 0) g_leaf_stamps[(kb) * 8 + (ph)] = (long long)__builtin_readcyclecounter(); } while (0)
#endif
#define LEAF_WAVE_END(kb, w) do { if ((threadIdx.x & 63) == 0) g_leaf_wave_end[(kb) * 4 + (w)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define LEAF_STAMP(kb, ph) do { } while (0)
#define LEAF_WAVE_END(kb, w) do { } while (0)
#endif

// Workgroup barrier / intra-wave LDS ordering that wait for LDS traffic only.  __syncthreads() and
// __builtin_amdgcn_fence() also wait for vmcnt(0), i.e. for the global stores of L issued just
// before -- a few thousand cycles per panel that nothing here depends on (the stores are outputs).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

constexpr int RL = 128;            // leaf size
constexpr int RPS = 18;            // LDS row stride of the 16-column panel (elements): conflict-free fragment reads
constexpr int RL_THREADS = 320;      // four waves that own the tiles + one wave that runs the pivot chains
constexpr int RL_TILE_THREADS = 256;

__device__ __forceinline__ double rl_readlane(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float rl_readlane(float v, int src_lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}
// Value of lane J of this lane's row of 16 lanes, on every lane of that row: one DPP move
// (v_mov_b64_dpp / v_mov_b32_dpp row_newbcast:J) -- no SGPR round trip, no LDS.
template <int J, typename R> __device__ __forceinline__ R rl_rowbcast(R x) {
  return __builtin_amdgcn_update_dpp(x, x, 0x150 + J, 0xf, 0xf, false);
}

__device__ __forceinline__ void rl_rsqrt_sqrt(double p, double& rinv, double& root) {
  double y = __builtin_amdgcn_rsq(p);
  const double hp = 0.5 * p;
  y = y * fma(-hp * y, y, 1.5);
#ifdef GPFIT_LEAF_TWO_NEWTON
  y = y * fma(-hp * y, y, 1.5);
#endif
  double d = p * y;
  d = fma(fma(-d, d, p), 0.5 * y, d);
  rinv = y;
  root = d;
}
__device__ __forceinline__ void rl_rsqrt_sqrt(float p, float& rinv, float& root) {
  float y = __builtin_amdgcn_rsqf(p);
  const float hp = 0.5f * p;
  y = y * fmaf(-hp * y, y, 1.5f);
  float d = p * y;
  d = fmaf(fmaf(-d, d, p), 0.5f * y, d);
  rinv = y;
  root = d;
}

// DPP-fused pieces of the pivot step, in inline assembly because the compiler only offers the row
// broadcast as a separate move (three instructions per column update instead of one; the pivot
// chain is VALU-issue bound).  Hazard: a VGPR written by a VALU instruction needs two wait states
// before a DPP instruction reads it, and hipcc pads nothing around inline assembly -- the s_nop 1
// in front of the first DPP read of a freshly written register provides them (asm volatile
// statements keep their order, so the later DPP reads of the same register are already safe).
//   p = lane J's x on every lane of the row
template <int J> __device__ __forceinline__ double dpp_bcast_nop(double x) {
  double r;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(J));
  return r;
}
template <int J> __device__ __forceinline__ float dpp_bcast_nop(float x) {
  float r;
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(J));
  return r;
}
//   acc += (lane J's a) * b
template <int J, bool NOP> __device__ __forceinline__ void dpp_fmac(double& acc, double a, double b) {
  if constexpr (NOP)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(J));
  else
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(J));
}
template <int J, bool NOP> __device__ __forceinline__ void dpp_fmac(float& acc, float a, float b) {
  if constexpr (NOP)
    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(J));
  else
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(J));
}

//   x += (lane J's x) * b   (same register as accumulator and as DPP source)
template <int J, bool NOP> __device__ __forceinline__ void dpp_fmac_self(double& x, double b) {
  if constexpr (NOP)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b), "n"(J));
  else
    asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b), "n"(J));
}
template <int J, bool NOP> __device__ __forceinline__ void dpp_fmac_self(float& x, float b) {
  if constexpr (NOP)
    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b), "n"(J));
  else
    asm volatile("v_fmac_f32_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b), "n"(J));
}

template <typename R, int K, int J> __device__ __forceinline__ void static_for_cols(R (&v)[16], R vk, R nvk);
template <typename R, int K, int C> __device__ __forceinline__ void static_for_inv(R (&yh)[4], R m);

// Pivot k of the 16 x 16 factorisation, then the rest; compile-time recursion because the DPP lane select is an
// immediate.  Lane 16 g + i holds row i of the block in v[0..15] -- all four DPP rows (g = 0..3) run the factor sweep
// redundantly: the lanes would idle otherwise and nothing has to cross a row of 16 lanes -- and columns
// {g, 4 + g, 8 + g, 12 + g} of row i of the inverse sweep in yh[0..3]: yh_i = e_i - sum_{k<i} l_ik y_k are the rows of
// L^-1 before their final division by l_ii; at pivot k the lanes below take  yh_i -= (l_ik / l_kk) yh_k  straight
// from lane k of their own DPP row (the broadcast index is the pivot row: the same for every column, whichever group
// holds it), columns 0 .. k only: floor(k / 4) + 1 instructions instead of k + 1 (40 instead of 136 per panel).
template <typename R, int K>
__device__ __forceinline__ void static_for_pivots(R (&v)[16], R (&yh)[4], R& myr, int& first_bad, int lane, int base) {
  if constexpr (K < 16) {
    R p = dpp_bcast_nop<K>(v[K]);
    const bool bad = !(p > (R)0);
    first_bad = (bad && first_bad == 0) ? (base + K + 1) : first_bad;
    p = bad ? (R)1 : p;
    R rinv, dkk;
    rl_rsqrt_sqrt(p, rinv, dkk);
    myr = (lane == K) ? rinv : myr;
    const R vk = (lane == K) ? dkk : v[K] * rinv;
    v[K] = vk;
    static_for_cols<R, K, K + 1>(v, vk, -vk);
    const R m = (lane > K) ? -vk * rinv : (R)0;
    static_for_inv<R, K, 0>(yh, m);
    static_for_pivots<R, K + 1>(v, yh, myr, first_bad, lane, base);
  }
}
template <typename R, int K, int J>
__device__ __forceinline__ void static_for_cols(R (&v)[16], R vk, R nvk) {
  if constexpr (J < 16) {
    dpp_fmac<J, J == K + 1>(v[J], vk, nvk);  // v[j] -= l_jk l_ik, l_jk = lane j's v[k]
    static_for_cols<R, K, J + 1>(v, vk, nvk);
  }
}
template <typename R, int K, int C>
__device__ __forceinline__ void static_for_inv(R (&yh)[4], R m) {
  if constexpr (4 * C <= K) {
    dpp_fmac_self<K, C == 0>(yh[C], m);      // yh_i[4 C + g] += m_i * (lane k's yh[C]);  columns beyond k hold zeros there
    static_for_inv<R, K, C + 1>(yh, m);
  }
}

// Column ownership of the four tile waves.  Tile column j has 8 - j tiles; a wave owns two whole columns (the inverse
// needs the tiles of a column together).  Wave 0 shares its SIMD with the pivot wave, whose VALU-bound chains outrank
// it (s_setprio): with nine tiles per wave it finished its update pass 2 k cycles behind the other three, so it
// owns the two shortest columns (3 tiles) and the other waves 11 tiles each.
constexpr int RL_SLOTS = 11;
__host__ __device__ constexpr int rl_col_a(int w) { return w == 0 ? 6 : w - 1; }          // 6, 0, 1, 2
__host__ __device__ constexpr int rl_col_b(int w) { return w == 0 ? 7 : 6 - w; }          // 7, 5, 4, 3
__host__ __device__ constexpr int rl_nslots(int w) { return (8 - rl_col_a(w)) + (8 - rl_col_b(w)); }   // 3, 11, 11, 11
__host__ __device__ constexpr int rl_slot_i(int w, int t) { return t < 8 - rl_col_a(w) ? rl_col_a(w) + t : rl_col_b(w) + (t - (8 - rl_col_a(w))); }
__host__ __device__ constexpr int rl_slot_j(int w, int t) { return t < 8 - rl_col_a(w) ? rl_col_a(w) : rl_col_b(w); }

// One tile column C of wave W in panel kb (slots T0 .. T0 + 7 - C hold rows C .. 7).  Either the whole column is
// still Schur complement (C > kb) or it is in its inverse phase (row kb becomes X, the rows below running sums):
// ONE wave-uniform branch per column and pass, the slots inside straight-line apart from the one special tile; every
// product accumulates in place.
template <typename R, int W, int C, int T0>
__device__ __forceinline__ void leaf_update_column(typename Real<R>::acc_t (&acc)[RL_SLOTS], const R* __restrict__ P,
                                                   const R (&dneg)[4], const typename Real<R>::acc_t& dvt, R* __restrict__ RawN,
                                                   R* __restrict__ Dg0, R* __restrict__ Dg1, int kb, const int (&kr)[4], int fr) {
  using Acc = typename Real<R>::acc_t;
  constexpr int NR = 8 - C;          // rows C .. 7
  const int nx = kb + 1;
  R fa[2][4];
  auto row_frag = [&](int i, R (&f)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) f[r] = P[(16 * i + fr) * RPS + kr[r]];          // rows <= kb: stale, never used
  };
  if (C > kb) {
    // Schur complement  S[i, C] -= L[i, kb] L[C, kb]^T: the column fragment negated once
    R nfb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) nfb[r] = -P[(16 * C + fr) * RPS + kr[r]];
    row_frag(C, fa[0]);
#pragma unroll
    for (int t = 0; t < NR; ++t) {
      const int i = C + t;
      Acc& a = acc[T0 + t];
      if (t + 1 < NR) row_frag(i + 1, fa[(t + 1) & 1]);
      const R (&f)[4] = fa[t & 1];
      if (!(t == 0 && C == nx)) {           // the diagonal tile of column kb + 1 was taken over by the pivot wave
#pragma unroll
        for (int r = 0; r < 4; ++r) a = Real<R>::mfma(f[r], nfb[r], a);
      }
      if (C == nx && t > 0) {
        // final through panel kb: a row of the raw column of the next panel
#pragma unroll
        for (int r = 0; r < 4; ++r) RawN[(16 * i + kr[r]) * RPS + fr] = a[r];
      } else if (C == nx + 1 && t == 0) {
        // the diagonal tile of column kb + 2, final through panel kb: into the mailbox of the pivot wave
        R* Dg = (C & 1) ? Dg1 : Dg0;
#pragma unroll
        for (int r = 0; r < 4; ++r) Dg[kr[r] * RPS + fr] = a[r];
      }
      __builtin_amdgcn_sched_barrier(0);    // (see leaf_update_pass)
    }
  } else {
    // inverse phase: X[kb, C] = -Dinv Y[kb, C]  (X[kb, kb] = Dinv), then  Y[i, C] += L[i, kb] X[kb, C]  for i > kb
    // (the row-kb slot is found by an unrolled search: a run-time register index is not an option)
    Acc xrow = acc_zero<R>();
#pragma unroll
    for (int t = 0; t < NR; ++t) {
      if (C + t == kb) {
        if (C == kb) {
          xrow = dvt;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) xrow = Real<R>::mfma(dneg[r], acc[T0 + t][r], xrow);
        }
        acc[T0 + t] = xrow;
      }
    }
    row_frag(C, fa[0]);
#pragma unroll
    for (int t = 0; t < NR; ++t) {
      const int i = C + t;
      Acc& a = acc[T0 + t];
      if (t + 1 < NR) row_frag(i + 1, fa[(t + 1) & 1]);
      const R (&f)[4] = fa[t & 1];
      if (i > kb) {
        if (C == kb) a = acc_zero<R>();     // column kb held the raw panel until now
#pragma unroll
        for (int r = 0; r < 4; ++r) a = Real<R>::mfma(f[r], xrow[r], a);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// Step (3) of a panel for tile wave W, specialised per wave: the tile coordinates of the slots are compile-time
// constants (LDS offsets become immediates, every guard compares the panel counter with a constant), while the panel
// loop around it stays rolled -- four copies of this pass instead of one generic copy whose guards and addresses are
// re-derived from the wave index for every slot and panel (measured on the generic copy: 3.8 k cycles per panel with
// only two active slots, 5.5-7.5 k with all of them, for at most 2.3 k cycles of matrix-pipe issue).  The row fragment
// L[i, kb] of the next slot is requested before the products of the current one; nothing moves across a slot boundary
// (sched_barrier): left to itself the scheduler hoists the fragment reads of ALL slots to the top of the pass (88 more
// registers live at once: spills).
template <typename R, int W>
__device__ __forceinline__ void leaf_update_pass(typename Real<R>::acc_t (&acc)[RL_SLOTS], const R* __restrict__ P,
                                                 const R* __restrict__ Dv, R* __restrict__ RawN, R* __restrict__ Dg0,
                                                 R* __restrict__ Dg1, int kb, const int (&kr)[4], int fr) {
  using Acc = typename Real<R>::acc_t;
  constexpr int CA = rl_col_a(W), CB = rl_col_b(W);
  R dneg[4];
  Acc dvt;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    dneg[r] = -Dv[fr * RPS + kr[r]];
    dvt[r] = Dv[kr[r] * RPS + fr];
  }
  leaf_update_column<R, W, CA, 0>(acc, P, dneg, dvt, RawN, Dg0, Dg1, kb, kr, fr);
  leaf_update_column<R, W, CB, 8 - CA>(acc, P, dneg, dvt, RawN, Dg0, Dg1, kb, kr, fr);
}

// Prologue / epilogue of tile wave W (its tiles in, the raw column 0 and the first two diagonal tiles into LDS; the
// finished rows of the inverse out): specialised per wave like the update pass.
template <typename R, int W>
__device__ __forceinline__ void leaf_load_tiles(typename Real<R>::acc_t (&acc)[RL_SLOTS], const R* __restrict__ A, int64_t lda,
                                                R* __restrict__ Raw0, R* __restrict__ Dg0, R* __restrict__ Dg1,
                                                const int (&kr)[4], int fr) {
#pragma unroll
  for (int t = 0; t < RL_SLOTS; ++t) {
    if (t >= rl_nslots(W)) { acc[t] = acc_zero<R>(); continue; }
    const int i = rl_slot_i(W, t), j = rl_slot_j(W, t);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = A[(int64_t)(16 * i + kr[r]) * lda + 16 * j + fr];
    // prologue of the pipeline: raw column 0 and the diagonal tiles of columns 0 and 1
    if (j == 0 && i > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) Raw0[(16 * i + kr[r]) * RPS + fr] = acc[t][r];
    }
    if (i == j && j <= 1) {
      R* Dg = j ? Dg1 : Dg0;
#pragma unroll
      for (int r = 0; r < 4; ++r) Dg[kr[r] * RPS + fr] = acc[t][r];
    }
  }
}
template <typename R, int W>
__device__ __forceinline__ void leaf_store_inverse(const typename Real<R>::acc_t (&acc)[RL_SLOTS], R* __restrict__ Linv,
                                                   int64_t ldi, const int (&kr)[4], int fr) {
#pragma unroll
  for (int t = 0; t < rl_nslots(W); ++t) {
    const int i = rl_slot_i(W, t), j = rl_slot_j(W, t);
#pragma unroll
    for (int r = 0; r < 4; ++r) Linv[(int64_t)(16 * i + kr[r]) * ldi + 16 * j + fr] = acc[t][r];
  }
}

// The panel loop is a real loop: the code is executed eight times and stays in the instruction cache.  (A fully
// unrolled, per-wave specialised version of this kernel -- 100 KiB of straight-line code -- spent 83 % of its wave
// cycles waiting for instruction fetches.)  Round 4: the passes that walk a wave's tile slots ARE specialised per wave
// (leaf_update_pass and the two helpers above, four copies each, selected by one switch on the wave index) while the
// panel loop stays rolled, and the pivot wave has a loop of its own: its path never touches the accumulators, so their
// 88 registers and the 64 of the pivot sweep (v, yh) overlay instead of adding up.
// One workgroup per block of the batch (LeafBatchT, kernels.h): blocks of several factorisations that
// have reached a leaf together (the K~ and V chains of a unit, the chains of several units) share the launch.
template <typename R>
__global__ __launch_bounds__(RL_THREADS, 2) void chol_leaf_reg_kernel(LeafBatchT<R> bt) {
  const R* __restrict__ A = bt.A[blockIdx.x];
  R* __restrict__ L = bt.L[blockIdx.x];
  R* __restrict__ Linv = bt.Li[blockIdx.x];
  int* __restrict__ info = bt.info[blockIdx.x];
  const int64_t lda = bt.lda, ldl = bt.ldl, ldi = bt.ldi;
  const int info_base = bt.info_base;
  using Acc = typename Real<R>::acc_t;
  using V = typename Real<R>::vec_t;
  constexpr int EPC = Real<R>::EPC;
  // LDS (48 KiB).  The raw column and the solved rows need one buffer each: raw column j is read before barrier
  // B2(j) (by (2) and by the pivot wave) and column j + 1 is written behind it; L[., j] is written before B2(j), read
  // behind it, and L[., j + 1] is written behind B1(j + 1).  Dinv and the mailbox alternate by panel parity (the
  // pivot wave writes Dinv of panel j + 1 while the tile waves still read that of panel j).
  __shared__ __attribute__((aligned(16))) R Rawb[1][RL * RPS];   // raw column j: S[i, j] for i > j, final through panel j - 1
  __shared__ __attribute__((aligned(16))) R Lpb[1][RL * RPS];    // L[i, j] for i > j (the solved rows of panel j)
  __shared__ __attribute__((aligned(16))) R Dbuf[2][16 * RPS];   // inverse of the 16 x 16 diagonal factor of panel j
  __shared__ __attribute__((aligned(16))) R Dgb[2][16 * RPS];    // mailbox: diagonal tile S[j, j], final through panel j - 2
  __shared__ __attribute__((aligned(16))) R Sx[16 * RPS];        // scratch of the pivot wave (layout changes)
  // Arrival counter of the barrier B2 (between the row solve (2) and the update pass (3) of a panel).  B2 is a
  // barrier of the four tile waves only: the pivot wave never reads what (2) writes, and made to wait for it (an
  // s_barrier is workgroup-wide) it idled 1 k cycles per panel at the head of the kernel's critical chain.  It still
  // ARRIVES, once its reads of the raw column are done -- the update pass overwrites that buffer -- but does not wait.
  __shared__ unsigned b2_arrivals;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15;
  int kr[4];  // k index (= C/D row) this lane carries in register r
#pragma unroll
  for (int r = 0; r < 4; ++r) kr[r] = Real<R>::crow(lane, r);

  if (tid == 0) b2_arrivals = 0u;          // (visible to everybody behind B0)
  auto b2_arrive = [&]() {                 // everything this wave has sent to LDS so far is done, then one count
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(&b2_arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto b2_wait = [&](unsigned target) {    // tile waves: until all five waves have arrived `target / 5` times
    while (__hip_atomic_load(&b2_arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) { }
    asm volatile("" ::: "memory");
  };
  if (wave == 4) {
    // ================= the pivot wave: the pivot chains, one panel ahead of the tile waves =================
    // the chains are the critical path of the kernel and VALU-issue bound; the wave shares its SIMD with a
    // tile wave whose updates would otherwise take every other issue slot (6.3 k instead of 3.9 k cycles per chain)
    __builtin_amdgcn_s_setprio(3);
    LEAF_STAMP(8, 0);
    // factor the 16 x 16 block held row-major in Sx (a row per lane) and invert the factor by the same sweep;
    // Dinv -> Dbuf[nx & 1], the rows of L -> memory
    auto pivot_block = [&](int nx, int lane_o) {
      const int li = lane_o & 15, lg = lane_o >> 4;     // row of the block / DPP row (column group of the inverse)
      R v[16], yh[4];
      {
        const R* row = Sx + li * RPS;
#pragma unroll
        for (int q = 0; q < 16 / EPC; ++q) {
          const V w = *reinterpret_cast<const V*>(row + EPC * q);
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[EPC * q + e] = w[e];
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) yh[c] = (4 * c + lg == li) ? (R)1 : (R)0;
      int first_bad = 0;
      R myr = (R)0;  // the lanes of row k keep 1 / L_kk
      LEAF_STAMP(nx, 1);
      static_for_pivots<R, 0>(v, yh, myr, first_bad, li, 16 * nx);
      LEAF_STAMP(nx, 2);
      if (lane == 0 && first_bad != 0) atomicCAS(info, 0, info_base + first_bad);
      {
        // Dinv: every lane its four columns (zero above the diagonal by construction); the diagonal rows of L are never
        // read from LDS -- only their inverse is -- and go to memory only (first DPP row)
        R* drow = Dbuf[nx & 1] + li * RPS + lg;
#pragma unroll
        for (int c = 0; c < 4; ++c) drow[4 * c] = yh[c] * myr;
      }
      if (lane < 16) {
        R* grow = L + (int64_t)(16 * nx + lane) * ldl + 16 * nx;
#pragma unroll
        for (int q = 0; q < 16 / EPC; ++q) {
          V lv;
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            const int j = EPC * q + e;
            lv[e] = (j <= lane) ? v[j] : (R)0;
          }
          *reinterpret_cast<V*>(grow + EPC * q) = lv;
        }
      }
      LEAF_STAMP(nx, 3);
    };
    {
      // column 0: nothing to update -- the wave fetches the diagonal tile itself and runs the first chain while the
      // tile waves are still loading theirs (their prologue is 9 k cycles, the chain 4 k)
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      if (lane < 16) {
        const R* arow = A + (int64_t)lane * lda;
        R* srow = Sx + lane * RPS;
#pragma unroll
        for (int q = 0; q < 16 / EPC; ++q) *reinterpret_cast<V*>(srow + EPC * q) = *reinterpret_cast<const V*>(arow + EPC * q);
      }
      lds_fence();
      pivot_block(0, lane_o);
    }
    lds_barrier();   // B0: raw column 0 and the diagonal tile of column 1 are in LDS (tile waves), Dinv of panel 0 (here)
#pragma unroll 1
    for (int kb = 0; kb < 8; ++kb) {
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      const int nx = kb + 1;
      const R* const Raw = Rawb[0];                        // raw column kb
      const R* const Dv = Dbuf[kb & 1];                    // Dinv of panel kb
      lds_barrier();   // B1(kb)
      LEAF_STAMP(kb, 4);
      if (nx < 8) {
        // L[nx, kb]^T = Dinv S[nx, kb]^T: the products and the order of the sums of L[nx, kb] = S[nx, kb] Dinv^T, which
        // the tile waves form in (2) (same bits), with the operands exchanged -- the C/D registers then hold
        // L^T[k, n] = L[n, k], which is the A fragment L[m, k] AND the B fragment L^T[k, n] of the next product
        // as they stand: no trip through LDS between the two
        Acc lt = acc_zero<R>();
#pragma unroll
        for (int r = 0; r < 4; ++r) lt = Real<R>::mfma(Dv[fr * RPS + kr[r]], Raw[(16 * nx + fr) * RPS + kr[r]], lt);
        // S[nx, nx] -= L[nx, kb] L[nx, kb]^T  on the mailbox copy (final through panel kb - 1)
        Acc sd;
#pragma unroll
        for (int r = 0; r < 4; ++r) sd[r] = Dgb[nx & 1][kr[r] * RPS + fr];
        b2_arrive();   // B2(kb): the raw column has been read (the MFMAs above consumed it); no wait
#pragma unroll
        for (int r = 0; r < 4; ++r) sd = Real<R>::mfma(-lt[r], lt[r], sd);
#pragma unroll
        for (int r = 0; r < 4; ++r) Sx[kr[r] * RPS + fr] = sd[r];     // row-major for the row-per-lane pivot sweep
        lds_fence();
      } else {
        b2_arrive();
      }
      LEAF_STAMP(kb, 5);
      if (nx < 8) {
        LEAF_STAMP(nx, 0);
        pivot_block(nx, lane_o);
      }
    }
    return;
  }

  // ================= the four tile waves =================
  Acc acc[RL_SLOTS];
  switch (wave) {
    case 0: leaf_load_tiles<R, 0>(acc, A, lda, Rawb[0], Dgb[0], Dgb[1], kr, fr); break;
    case 1: leaf_load_tiles<R, 1>(acc, A, lda, Rawb[0], Dgb[0], Dgb[1], kr, fr); break;
    case 2: leaf_load_tiles<R, 2>(acc, A, lda, Rawb[0], Dgb[0], Dgb[1], kr, fr); break;
    default: leaf_load_tiles<R, 3>(acc, A, lda, Rawb[0], Dgb[0], Dgb[1], kr, fr); break;
  }
  lds_barrier();   // B0

  // Software pipeline over the panels with a dedicated pivot wave, synchronised by workgroup barriers only.
  // Panel j = 0 .. 7, after barrier B1(j) (Dinv of panel j, raw column j and the diagonal tile of column j + 1,
  // final through panel j - 1, are in LDS):
  //   tile waves:  (2) the rows below, L[i, j] = S[i, j] Dinv^T          [B2(j)]
  //                (3) update their tiles from panel j; the owner of column j + 1 puts its rows below the diagonal
  //                    into the raw buffer of panel j + 1, the owner of column j + 2 its diagonal tile (final
  //                    through panel j) into the mailbox                                         [B1(j + 1)]
  //   pivot wave:  its OWN copy of L[j + 1, j] from the raw column and Dinv (not waiting for (2)), the last Schur
  //                update of the diagonal tile of column j + 1 from it                          [B2(j)]
  //                the pivot chain of panel j + 1 -- a quarter of a panel's time, one wave's work -- beside (3)
  //                                                                                              [B1(j + 1)]
  // so the chain  Dinv(j) -> L[j + 1, j] -> S[j + 1, j + 1] -> pivots(j + 1)  never leaves the pivot wave and waits
  // for nobody; two barriers per panel, no flags.
#pragma unroll 1
  for (int kb = 0; kb < 8; ++kb) {
    // Left alone, the compiler hoists every LDS offset and store address of every slot of all four specialised
    // passes out of this loop (256 VGPRs and spills).  Re-deriving the lane's coordinates from an opaque copy of the
    // lane index each iteration keeps them inside: one and / shift per use.
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int fr = lane_o & 15;
    int kr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) kr[r] = Real<R>::crow(lane_o, r);
    const R* const Raw = Rawb[0];                        // raw column kb
    R* const P = Lpb[0];                                 // L rows of panel kb
    const R* const Dv = Dbuf[kb & 1];                    // Dinv of panel kb
    R* const RawN = Rawb[0];                             // raw column kb + 1 (assembled behind B2(kb))

    lds_barrier();   // B1(kb)
    // ---- (2) rows below: L[i, kb] = S[i, kb] Dinv^T, tiles kb+1 .. 7 dealt to the tile waves
    {
      Acc s0 = acc_zero<R>(), s1 = acc_zero<R>();
      const int i0 = kb + 1 + wave, i1 = kb + 5 + wave;
      // both products unconditionally (rows clamped into the panel; results of rows that do not exist are simply
      // not stored): straight-line code, so the two dependent MFMA chains and their LDS reads interleave
      const int c0 = i0 < 8 ? i0 : 7, c1 = i1 < 8 ? i1 : 7;
      {
        R a0[4], a1[4], dd[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          a0[r] = Raw[(16 * c0 + fr) * RPS + kr[r]];
          a1[r] = Raw[(16 * c1 + fr) * RPS + kr[r]];
          dd[r] = Dv[fr * RPS + kr[r]];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s0 = Real<R>::mfma(a0[r], dd[r], s0);
          s1 = Real<R>::mfma(a1[r], dd[r], s1);
        }
      }
      if (i0 < 8) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          P[(16 * i0 + kr[r]) * RPS + fr] = s0[r];
          L[(int64_t)(16 * i0 + kr[r]) * ldl + 16 * kb + fr] = s0[r];
        }
      }
      if (i1 < 8) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          P[(16 * i1 + kr[r]) * RPS + fr] = s1[r];
          L[(int64_t)(16 * i1 + kr[r]) * ldl + 16 * kb + fr] = s1[r];
        }
      }
    }
    b2_arrive();
    b2_wait(5u * (unsigned)(kb + 1));   // B2(kb): L[., kb] is in LDS, and the pivot wave is done with the raw column
    // ---- (3) every tile wave updates its tiles from panel kb (leaf_update_pass, one specialisation per wave)
    switch (wave) {
      case 0: leaf_update_pass<R, 0>(acc, P, Dv, RawN, Dgb[0], Dgb[1], kb, kr, fr); break;
      case 1: leaf_update_pass<R, 1>(acc, P, Dv, RawN, Dgb[0], Dgb[1], kb, kr, fr); break;
      case 2: leaf_update_pass<R, 2>(acc, P, Dv, RawN, Dgb[0], Dgb[1], kb, kr, fr); break;
      default: leaf_update_pass<R, 3>(acc, P, Dv, RawN, Dgb[0], Dgb[1], kb, kr, fr); break;
    }
    if (wave == 0) {
      // wave 0 owns three tiles and is done with its pass 1.5 k cycles before the others: it writes the zeros of the
      // strict upper tiles of tile row kb of both outputs (the callers read whole 128-blocks), 28 tiles over the panels
      const int row = 16 * kb + (lane_o >> 2), cq = 4 * (lane_o & 3);
      V zero2;
#pragma unroll
      for (int e = 0; e < EPC; ++e) zero2[e] = (R)0;
      for (int tj = kb + 1; tj < 8; ++tj) {
        R* lrow = L + (int64_t)row * ldl + 16 * tj + cq;
        R* irow = Linv + (int64_t)row * ldi + 16 * tj + cq;
#pragma unroll
        for (int q = 0; q < 4 / EPC; ++q) {
          *reinterpret_cast<V*>(lrow + EPC * q) = zero2;
          *reinterpret_cast<V*>(irow + EPC * q) = zero2;
        }
      }
      LEAF_STAMP(kb, 6);
    }
    LEAF_WAVE_END(kb, wave);
  }

  // all rows of the inverse are final
  switch (wave) {
    case 0: leaf_store_inverse<R, 0>(acc, Linv, ldi, kr, fr); break;
    case 1: leaf_store_inverse<R, 1>(acc, Linv, ldi, kr, fr); break;
    case 2: leaf_store_inverse<R, 2>(acc, Linv, ldi, kr, fr); break;
    default: leaf_store_inverse<R, 3>(acc, Linv, ldi, kr, fr); break;
  }
  LEAF_STAMP(8, 1);
}

template <typename R>
int launch_chol_leaf_batch(const LeafBatchT<R>& bt, hipStream_t s) {
  if (bt.n <= 0 || bt.n > GEMM_MAXB) {
    set_error("launch_chol_leaf_batch: 1 .. GEMM_MAXB blocks per launch");
    return -3;
  }
  hipLaunchKernelGGL(chol_leaf_reg_kernel<R>, dim3(bt.n), dim3(RL_THREADS), 0, s, bt);
  GP_HIP(hipGetLastError());
  return 0;
}

template <typename R>
int launch_chol_leaf_reg(const R* A, int64_t lda, R* L, int64_t ldl, R* Linv, int64_t ldi, int* info, int info_base,
                         hipStream_t s) {
  LeafBatchT<R> bt{};
  bt.n = 1; bt.A[0] = A; bt.L[0] = L; bt.Li[0] = Linv; bt.info[0] = info;
  bt.lda = lda; bt.ldl = ldl; bt.ldi = ldi; bt.info_base = info_base;
  return launch_chol_leaf_batch(bt, s);
}

template int launch_chol_leaf_batch<double>(const LeafBatchT<double>&, hipStream_t);
template int launch_chol_leaf_batch<float>(const LeafBatchT<float>&, hipStream_t);
template int launch_chol_leaf_reg<double>(const double*, int64_t, double*, int64_t, double*, int64_t, int*, int,
                                          hipStream_t);
template int launch_chol_leaf_reg<float>(const float*, int64_t, float*, int64_t, float*, int64_t, int*, int, hipStream_t);

}  // namespace gpfit
