// Cholesky leaf, register-resident variant: factor one 128 x 128 diagonal block and invert the
// factor with the matrix held in the accumulator registers of four waves and only the current
// 16-column panel in LDS (21 KiB instead of the 133 KiB of chol_leaf.hip).
//
// Why: the leaf sits on the critical path of both factorisation chains 2N/128 times per fit, and
// the two chains run concurrently on two streams.  A leaf that needs 133 KiB of LDS can only start
// on a CU with no other workgroup, so while the other chain (or a look-ahead GEMM of the same
// chain) keeps every CU busy with two 64 KiB GEMM workgroups the leaf waits for a whole CU to
// drain.  With <= 32 KiB of LDS and <= 128 VGPRs this kernel fits beside two such workgroups on
// any CU (2 x 64 + 32 = 160 KiB; 2 x 192 + 128 = 512 VGPRs per SIMD lane) and starts at once.
//
// Layout.  The block is cut into 8 x 8 tiles of 16 x 16; tile (i, j), i >= j, lives in the MFMA
// C/D layout (4 values per lane) in the registers of the wave that owns tile COLUMN j: wave w owns
// columns w and 7 - w (9 tiles each).  Column ownership is what makes the in-place inverse
// possible: for the fp64 16x16x4 MFMA the C/D register r of a tile holds exactly the rows
// k_r(lane) = (lane >> 4) + 4 r that the B operand of sub-step r needs (fp32: 4 (lane >> 4) + r, the
// sum over k simply runs in that order), so a register tile X[k, j] is used directly as the B
// operand of  Y[i, j] += L[i, k] X[k, j]  -- no lane movement -- as long as the same wave holds
// both, i.e. as long as tiles of one column stay together.
//
// Right-looking over the eight 16-column panels kb:
//   (1) the owner of column kb writes its tiles S[kb.., kb] (the Schur complement so far) to the
//       LDS panel P[128][16] and factors the 16 x 16 diagonal block with a row per lane, pivots
//       and multipliers moving between lanes with v_readlane; it also inverts that factor
//       (Dinv, a column per lane) -- the panel solve below is then an MFMA product;
//   (2) the rows below the diagonal block:  L[i, kb] = S[i, kb] Dinv^T,  one tile per wave and
//       pass, written back to P and to global memory;
//   (3) every wave updates its own tiles from the panel:
//         columns j > kb (still Schur complement):  S[i, j] -= L[i, kb] L[j, kb]^T
//         columns j <= kb (already inverse):        X[kb, j] = -Dinv Y[kb, j]   (X[kb, kb] = Dinv)
//                                                   Y[i, j] += L[i, kb] X[kb, j]   for i > kb
//       so the registers of column j hold S[., j] until panel j has been factored and the rows of
//       L^-1 (finished rows X, running sums Y) afterwards: the inverse costs no extra storage and
//       is complete when the last panel is.
// info: LAPACK-style, as chol_leaf.hip (first non-positive pivot, offending pivot replaced by 1).
#include "gemm_core.h"
#include "kernels.h"

#include <cstdlib>

namespace gpfit {

// scripts/scratch/dev_leaf_time.hip defines GPFIT_LEAF_STAMPS and includes this file: shader-clock stamps of
// wave `owner` / wave 0 at the phase boundaries of every panel (never compiled into the library)
#ifdef GPFIT_LEAF_STAMPS
__device__ long long g_leaf_stamps[9 * 8];
#define LEAF_STAMP(kb, ph) do { if ((threadIdx.x & 63) == 0) g_leaf_stamps[(kb) * 8 + (ph)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define LEAF_STAMP(kb, ph) do { } while (0)
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
  y = y * fma(-hp * y, y, 1.5);
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
template <typename R, int K, int J> __device__ __forceinline__ void static_for_inv(R (&yh)[16], R m);

// Pivot k of the 16 x 16 factorisation (row i of the block in lane i), then the rest; compile-time
// recursion because the DPP lane select is an immediate.  The inverse of the factor is built by the
// same sweep: yh_i = e_i - sum_{k<i} l_ik y_k are the rows of L^-1 before their final division
// by l_ii; at pivot k the lanes below take  yh_i -= (l_ik / l_kk) yh_k  (columns 0..k) straight
// from lane k by DPP, which fills issue slots the rsqrt chain of the next pivot leaves empty.
template <typename R, int K>
__device__ __forceinline__ void static_for_pivots(R (&v)[16], R (&yh)[16], R& myr, int& first_bad, int lane, int base) {
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
template <typename R, int K, int J>
__device__ __forceinline__ void static_for_inv(R (&yh)[16], R m) {
  if constexpr (J <= K) {
    dpp_fmac_self<K, J == 0>(yh[J], m);      // yh_i[j] += m_i * (lane k's yh[j])
    static_for_inv<R, K, J + 1>(yh, m);
  }
}

// The panel loop is a real loop and the wave index a run-time value: the code is executed eight
// times and stays in the instruction cache.  (A fully unrolled, per-wave specialised version of
// this kernel -- 100 KiB of straight-line code -- spent 83 % of its wave cycles waiting for
// instruction fetches: 3.8 k instructions per wave in 150 k cycles.)  Register indices stay static
// because the nine tile slots of a wave are walked by an unrolled loop whose body branches on
// wave-uniform run-time conditions, and the one register operand that would need a run-time index
// -- X[kb, j] as the B operand of the Y updates -- is copied to a fixed tile (xrow) when it is
// produced: the slots of a column are walked in row order, so its row-kb slot comes first.
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
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool pivot_wave = wave == 4;
  const int fr = lane & 15;
  int kr[4];  // k index (= C/D row) this lane carries in register r
#pragma unroll
  for (int r = 0; r < 4; ++r) kr[r] = Real<R>::crow(lane, r);
  // slot t of wave w: t < n1 -> tile (w + t, w), else tile (7 - w + t - n1, 7 - w)
  const int tw = pivot_wave ? 0 : wave;   // (the pivot wave owns no tiles; its slots are never touched)
  const int n1w = 8 - tw;
  auto slot_iw = [&](int t) { return t < n1w ? tw + t : (7 - tw) + (t - n1w); };
  auto slot_jw = [&](int t) { return t < n1w ? tw : 7 - tw; };

  Acc acc[9];
  if (!pivot_wave) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int i = slot_iw(t), j = slot_jw(t);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = A[(int64_t)(16 * i + kr[r]) * lda + 16 * j + fr];
      // prologue of the pipeline: raw column 0 and the diagonal tiles of columns 0 and 1
      if (j == 0 && i > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Rawb[0][(16 * i + kr[r]) * RPS + fr] = acc[t][r];
      }
      if (i == j && j <= 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Dgb[j][kr[r] * RPS + fr] = acc[t][r];
      }
    }
    // strict upper tiles of both outputs are zero (the callers read whole 128-blocks)
    for (int e = tid; e < 28 * 64; e += RL_TILE_THREADS) {
      const int tix = e >> 6, q = e & 63;
      int ti = 0, rem = tix;  // tix -> (ti, tj) with tj > ti: rows 0..6 hold 7, 6, .. 1 tiles
      while (rem >= 7 - ti) { rem -= 7 - ti; ++ti; }
      const int tj = ti + 1 + rem;
      const int row = 16 * ti + (q >> 2), col = 16 * tj + 4 * (q & 3);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        L[(int64_t)row * ldl + col + c] = (R)0;
        Linv[(int64_t)row * ldi + col + c] = (R)0;
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = acc_zero<R>();
    // the pivot chains are the critical path of the kernel and VALU-issue bound; the wave shares its SIMD with a
    // tile wave whose updates would otherwise take every other issue slot (6.3 k instead of 3.9 k cycles per chain)
    __builtin_amdgcn_s_setprio(3);
  }
  LEAF_STAMP(8, 0);

  // The pivot wave's part of a panel: factor the 16 x 16 block held row-major in Sx (a row per lane) and invert the
  // factor by the same sweep; Dinv -> Dbuf[nx & 1], the rows of L -> memory.
  auto pivot_block = [&](int nx, int lane_o) {
    R v[16], yh[16];
    {
      const R* row = Sx + (lane & 15) * RPS;
#pragma unroll
      for (int q = 0; q < 16 / EPC; ++q) {
        const V w = *reinterpret_cast<const V*>(row + EPC * q);
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[EPC * q + e] = (lane < 16) ? w[e] : (R)0;
      }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) yh[j] = (j == lane_o) ? (R)1 : (R)0;
    int first_bad = 0;
    R myr = (R)0;  // lane k keeps 1 / L_kk
    LEAF_STAMP(nx, 1);
    static_for_pivots<R, 0>(v, yh, myr, first_bad, lane, 16 * nx);
    LEAF_STAMP(nx, 2);
    if (lane == 0 && first_bad != 0) atomicCAS(info, 0, info_base + first_bad);
    if (lane < 16) {
      R* drow = Dbuf[nx & 1] + lane * RPS;
      R* grow = L + (int64_t)(16 * nx + lane) * ldl + 16 * nx;
#pragma unroll
      for (int q = 0; q < 16 / EPC; ++q) {
        V lv, dv;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const int j = EPC * q + e;
          lv[e] = (j <= lane) ? v[j] : (R)0;
          dv[e] = yh[j] * myr;                       // zero above the diagonal by construction
        }
        *reinterpret_cast<V*>(drow + EPC * q) = dv;      // (the diagonal rows of L are never read from LDS --
        *reinterpret_cast<V*>(grow + EPC * q) = lv;      //  only their inverse is -- and go to memory only)
      }
    }
    LEAF_STAMP(nx, 3);
  };

  lds_barrier();   // B0: raw column 0 and the first two diagonal tiles are in LDS
  if (pivot_wave) {
    // column 0: nothing to update, the diagonal tile goes from the mailbox to the row-per-lane layout
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
#pragma unroll
    for (int r = 0; r < 4; ++r) Sx[kr[r] * RPS + fr] = Dgb[0][kr[r] * RPS + fr];
    lds_fence();
    pivot_block(0, lane_o);
  }

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
  Acc xrow = acc_zero<R>();  // X[kb, j] of the column being walked (its row-kb slot comes before its later rows)
#pragma unroll 1
  for (int kb = 0; kb < 8; ++kb) {
    // The tile coordinates of the slots depend on the wave only; left alone, the compiler hoists every
    // LDS offset and store address of every slot out of this loop (100+ VGPRs, spilled to scratch).
    // Re-deriving them from an opaque copy of the wave index each iteration costs one add per access.
    int wv = tw, lane_o = lane;
    asm volatile("" : "+s"(wv));
    asm volatile("" : "+v"(lane_o));
    const int n1 = 8 - wv;
    auto slot_i = [&](int t) { return t < n1 ? wv + t : (7 - wv) + (t - n1); };
    auto slot_j = [&](int t) { return t < n1 ? wv : 7 - wv; };
    const int nx = kb + 1;
    const R* const Raw = Rawb[0];                        // raw column kb
    R* const P = Lpb[0];                                 // L rows of panel kb
    const R* const Dv = Dbuf[kb & 1];                    // Dinv of panel kb
    R* const RawN = Rawb[0];                             // raw column kb + 1 (assembled behind B2(kb))

    lds_barrier();   // B1(kb)
    if (pivot_wave) {
      LEAF_STAMP(kb, 4);
      if (nx < 8) {
        // L[nx, kb] = S[nx, kb] Dinv^T  (the same product the tile waves form in (2): same bits)
        Acc lt = acc_zero<R>();
#pragma unroll
        for (int r = 0; r < 4; ++r) lt = Real<R>::mfma(Raw[(16 * nx + fr) * RPS + kr[r]], Dv[fr * RPS + kr[r]], lt);
#pragma unroll
        for (int r = 0; r < 4; ++r) Sx[kr[r] * RPS + fr] = lt[r];
        lds_fence();
        // S[nx, nx] -= L[nx, kb] L[nx, kb]^T  on the mailbox copy (final through panel kb - 1)
        Acc sd;
        R f[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          f[r] = Sx[fr * RPS + kr[r]];
          sd[r] = Dgb[nx & 1][kr[r] * RPS + fr];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sd = Real<R>::mfma(-f[r], f[r], sd);
#pragma unroll
        for (int r = 0; r < 4; ++r) Sx[kr[r] * RPS + fr] = sd[r];     // row-major for the row-per-lane pivot sweep
        lds_fence();
      }
      lds_barrier();   // B2(kb)
      LEAF_STAMP(kb, 5);
      if (nx < 8) {
        LEAF_STAMP(nx, 0);
        pivot_block(nx, lane_o);
      }
      continue;
    }

    // ---- tile waves ----
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
    lds_barrier();   // B2(kb): L[., kb] is in LDS
    // ---- (3) every tile wave updates its tiles from panel kb
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int i = slot_i(t), j = slot_j(t);
      Acc& a = acc[t];
      if (j > kb) {
        if (i == nx && j == nx) continue;   // the pivot wave took this tile over (mailbox of the previous panel)
        // Schur complement  S[i, j] -= L[i, kb] L[j, kb]^T
#pragma unroll
        for (int r = 0; r < 4; ++r)
          a = Real<R>::mfma(-P[(16 * i + fr) * RPS + kr[r]], P[(16 * j + fr) * RPS + kr[r]], a);
        if (j == nx) {
          // final through panel kb: a row of the raw column of the next panel
#pragma unroll
          for (int r = 0; r < 4; ++r) RawN[(16 * i + kr[r]) * RPS + fr] = a[r];
        } else if (i == nx + 1 && j == nx + 1) {
          // the diagonal tile of column kb + 2, final through panel kb: into the mailbox of the pivot wave
#pragma unroll
          for (int r = 0; r < 4; ++r) Dgb[j & 1][kr[r] * RPS + fr] = a[r];
        }
      } else if (i == kb) {
        // row kb of the inverse:  X[kb, kb] = Dinv,  X[kb, j] = -Dinv Y[kb, j]
        Acc nxv;
        if (j == kb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) nxv[r] = Dv[kr[r] * RPS + fr];
        } else {
          nxv = acc_zero<R>();
#pragma unroll
          for (int r = 0; r < 4; ++r) nxv = Real<R>::mfma(-Dv[fr * RPS + kr[r]], a[r], nxv);
        }
        a = nxv;
        xrow = nxv;
      } else if (i > kb) {
        // running sums of the rows still to come:  Y[i, j] += L[i, kb] X[kb, j]
        Acc y = (j == kb) ? acc_zero<R>() : a;  // column kb held the raw panel until now
#pragma unroll
        for (int r = 0; r < 4; ++r) y = Real<R>::mfma(P[(16 * i + fr) * RPS + kr[r]], xrow[r], y);
        a = y;
      }
    }
    if (wave == 0) LEAF_STAMP(kb, 6);
  }

  // all rows of the inverse are final (store addresses derived here, not kept live through the loop)
  if (pivot_wave) return;
  int wv2 = wave;
  asm volatile("" : "+s"(wv2));
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int i = t < 8 - wv2 ? wv2 + t : (7 - wv2) + (t - (8 - wv2)), j = t < 8 - wv2 ? wv2 : 7 - wv2;
#pragma unroll
    for (int r = 0; r < 4; ++r) Linv[(int64_t)(16 * i + kr[r]) * ldi + 16 * j + fr] = acc[t][r];
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
