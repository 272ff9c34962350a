// MFMA GEMM main loop for gfx950 (CDNA4), shared by the plain GEMM, the stream-K GEMM and the
// fused arc-cosine Gram kernel; fp64 (v_mfma_f64_16x16x4_f64) and fp32 (v_mfma_f32_16x16x4_f32)
// through the Real<> traits.  Internal header.
//
// Block = 256 threads = 4 waves (2 x 2), block tile T x T (T = 128 for large problems; 64 / 32
// so that small panels of the recursive Cholesky still spread over many CUs), K step of
// 16 (fp64) / 32 (fp32) = 128 bytes per operand row, wave tile T/2 x T/2 of 16 x 16 x 4 MFMA
// tiles (4 x 4 accumulators at T = 128), two LDS buffers, one barrier per K step.
//
// Full-tile instances stage operands global -> LDS by DMA (buffer_load_dwordx4 ... lds), with
// the bank-conflict permutation applied on the source address; MFMA fragments are double-buffered
// in registers so every LDS wait sits behind 16 MFMAs (gemm_mainloop below: 189 VGPRs at T = 128,
// 2 blocks / CU; plain 8192^3 fp64 at 72-73 TFLOP/s = 0.92-0.94 of the matrix peak).  Ragged
// (EDGE) instances keep a register-staged loop with swizzled k-major images (zero fill per row).
#pragma once
#include "common.h"

#include <cstdint>

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
// The address is split into a wave-uniform base (tile origin, advanced by a scalar add per K
// step) and per-thread byte offsets computed once before the loop, so that a K step costs one
// global_load (SGPR base + VGPR offset) per chunk and no address arithmetic.
template <typename R, bool KMAJOR, bool EDGE, int T> struct TileSrc {
  const char* base;     // uniform: &P[(k0, x0)]
  int64_t step;         // bytes per K step
  uint32_t off[T / 32];  // per-thread byte offsets of the chunks within the tile
  uint32_t valid;        // EDGE: bit i = chunk i inside the matrix

  __device__ __forceinline__ void init(const R* __restrict__ P, int64_t ld, int x0, int k0, int X, int tid) {
    constexpr int EPC = Real<R>::EPC, KT = Real<R>::KT;
    base = reinterpret_cast<const char*>(KMAJOR ? P + (int64_t)k0 * ld + x0 : P + (int64_t)x0 * ld + k0);
    step = (KMAJOR ? (int64_t)KT * ld : (int64_t)KT) * (int64_t)sizeof(R);
    valid = 0;
#pragma unroll
    for (int i = 0; i < T / 32; ++i) {
      const int c = tid + GEMM_THREADS * i;
      int x, k;
      if (KMAJOR) { k = c / (T / EPC); x = EPC * (c % (T / EPC)); }
      else { x = c >> 3; k = EPC * (c & 7); }
      off[i] = (uint32_t)((KMAJOR ? (int64_t)k * ld + x : (int64_t)x * ld + k) * (int64_t)sizeof(R));
      if (!EDGE || x0 + x < X) valid |= 1u << i;
    }
  }
  __device__ __forceinline__ void load(typename Real<R>::vec_t (&r)[T / 32]) const {
    using V = typename Real<R>::vec_t;
#pragma unroll
    for (int i = 0; i < T / 32; ++i) {
      if (!EDGE || ((valid >> i) & 1u)) r[i] = *reinterpret_cast<const V*>(base + off[i]);
      else r[i] = V{};
    }
  }
  __device__ __forceinline__ void advance() { base += step; }
};

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
// ---------------------------------------------------------------------------------------------
// Register-staged main loop (EDGE instances only: ragged M / N need the per-row zero fill).
// acc[mi][ni] += op(A)[row0.., k] * op(B)[k, col0..] over k in [kbeg, kend) (multiples of KT).
// smem: 4 * KT * T elements: [buf][A|B][KT][T] swizzled k-major images.
template <typename R, bool A_KMAJOR, bool B_KMAJOR, int T>
__device__ __forceinline__ void gemm_mainloop_regstage(const R* __restrict__ A, int64_t lda, const R* __restrict__ B,
                                                       int64_t ldb, int M, int N, int row0, int col0, int kbeg,
                                                       int kend, R* smem, typename Real<R>::acc_t (&acc)[T / 32][T / 32]) {
  using V = typename Real<R>::vec_t;
  constexpr int MI = T / 32, WT = T / 2, KT = Real<R>::KT, LT = KT * T;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;
  V ra[MI], rb[MI];

  if (kbeg >= kend) return;
  TileSrc<R, A_KMAJOR, true, T> sa;
  TileSrc<R, B_KMAJOR, true, T> sb;
  sa.init(A, lda, row0, kbeg, M, tid);
  sb.init(B, ldb, col0, kbeg, N, tid);
  sa.load(ra);
  sb.load(rb);
  __syncthreads();
  tile_sstore<R, A_KMAJOR, T>(ra, smem, tid);
  tile_sstore<R, B_KMAJOR, T>(rb, smem + LT, tid);
  __syncthreads();
  if (kbeg + KT < kend) {
    sa.advance();
    sb.advance();
    sa.load(ra);
    sb.load(rb);
  }
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += KT) {
    const bool more = (k0 + KT) < kend;
    const bool more2 = (k0 + 2 * KT) < kend;
    const R* As = smem + buf * 2 * LT;
    const R* Bs = As + LT;
    R* Sn = smem + (buf ^ 1) * 2 * LT;
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
      if (kk == KT / 8 && more) {
        tile_sstore<R, A_KMAJOR, T>(ra, Sn, tid);
        tile_sstore<R, B_KMAJOR, T>(rb, Sn + LT, tid);
        if (more2) {
          sa.advance();
          sb.advance();
          sa.load(ra);
          sb.load(rb);
        }
      }
    }
    __syncthreads();
    buf ^= 1;
  }
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA main loop (all full-tile instances).  Operand tiles go global -> LDS directly
// (buffer_load_dwordx4 ... lds: 64 lanes x 16 B = 1 KiB of LDS per wave-instruction, written
// lane-linearly), so the staging costs no VGPRs and no ds_write pass.  The LDS image of an operand
// depends on how its source is laid out (a DMA cannot transpose):
//   k-major source   ([k][x], x contiguous): image [KT][T], column index XOR 16*(k & SWK)
//       (the rows of one MFMA fragment read, 1 KiB apart, land in different bank groups);
//   k-contiguous source ([x][k]):            image [T][KT] (one 128-byte row per x), the eight
//       16-byte chunks of a row XOR-permuted by (x >> 1) & 7 (16 rows of a fragment read hit 16
//       different bank groups).
// Both permutations are applied on the SOURCE address of the DMA (per-lane) and again on the
// fragment read; the LDS write itself stays linear.
template <typename R, int T> struct LdsImage {
  static constexpr int EPC = Real<R>::EPC, KT = Real<R>::KT;
  static constexpr int SWK = (EPC == 2 ? 1 : 3) & (T / 16 - 1);
  // element offset of (x, k) in the image
  template <bool KMAJOR> static __device__ __forceinline__ int at(int x, int k) {
    if (KMAJOR) return k * T + (x ^ (16 * (k & SWK)));
    return x * KT + ((((k / EPC) ^ ((x >> 1) & 7))) * EPC) + (k % EPC);
  }
};

template <typename R, bool KMAJOR, int T> struct TileDma {
  static constexpr int MI = T / 32, EPC = Real<R>::EPC, KT = Real<R>::KT;
  const char* base;   // wave-uniform: &P[(k0, x0)]
  int64_t step;       // bytes per K step
  uint32_t voff[MI];  // per-lane source byte offsets of this wave's MI instructions
  int q0;             // first instruction index of this wave (LDS destination = q * 1 KiB)

  __device__ __forceinline__ void init(const R* __restrict__ P, int64_t ld, int x0, int k0, int wave, int lane) {
    base = reinterpret_cast<const char*>(KMAJOR ? P + (int64_t)k0 * ld + x0 : P + (int64_t)x0 * ld + k0);
    step = (KMAJOR ? (int64_t)KT * ld : (int64_t)KT) * (int64_t)sizeof(R);
    q0 = wave * MI;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int pch = (q0 + i) * 64 + lane;  // 16-byte chunk index inside the image
      int64_t e;
      if (KMAJOR) {
        constexpr int CPR = T / EPC;
        const int k = pch / CPR, xs = (pch % CPR) * EPC;
        e = (int64_t)k * ld + (xs ^ (16 * (k & LdsImage<R, T>::SWK)));
      } else {
        const int x = pch >> 3, j = (pch & 7) ^ ((x >> 1) & 7);
        e = (int64_t)x * ld + j * EPC;
      }
      voff[i] = (uint32_t)(e * (int64_t)sizeof(R));
    }
  }
  // image: LDS byte address of the operand image this tile goes to (wave-uniform)
  __device__ __forceinline__ void issue(uint32_t image) const {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, -1, 0x00020000);
#pragma unroll
    for (int i = 0; i < MI; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rs, (__attribute__((address_space(3))) void*)(uintptr_t)(image + (uint32_t)(q0 + i) * 1024u), 16, voff[i], 0, 0, 0);
  }
  // one of the MI pieces on its own (main loops that spread the pieces of a tile over the sub-steps of a K step)
  __device__ __forceinline__ void issue_one(uint32_t image, int i) const {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, -1, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(
        rs, (__attribute__((address_space(3))) void*)(uintptr_t)(image + (uint32_t)(q0 + i) * 1024u), 16, voff[i], 0, 0, 0);
  }
  __device__ __forceinline__ void advance() { base += step; }
};

// Wait until at most N of this wave's DMA instructions are outstanding and all its LDS reads have
// returned, then the workgroup barrier (a raw s_barrier: __syncthreads() would drain every DMA).
template <int N> __device__ __forceinline__ void dma_wait_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// ABL: ablation bits for scripts/scratch/dev_gemm_abl.hip only (1 no barrier, 2 no DMA in the loop);
// product code uses 0.
// NS = LDS stages (each one A tile + one B tile); the DMA runs NS-1 tiles ahead of the MFMAs.
// NS = 2 for the 128-tile (two co-resident workgroups hide each other's latency); the small-tile
// instances, used exactly when a launch cannot fill the chip, run one workgroup per CU and need
// several tiles in flight to cover the L2 / fabric latency (NS = 4 at T = 64, 8 at T = 32).
// smem: 2 * NS * KT * T elements.
template <typename R, bool A_KMAJOR, bool B_KMAJOR, bool EDGE, int T, int ABL = 0, int NS = 2>
__device__ __forceinline__ void gemm_mainloop(const R* __restrict__ A, int64_t lda, const R* __restrict__ B,
                                              int64_t ldb, int M, int N, int row0, int col0, int kbeg, int kend,
                                              R* smem, typename Real<R>::acc_t (&acc)[T / 32][T / 32],
                                              bool kdesc = false) {
  // kdesc: walk k from kend down to kbeg, for tiles whose ranges END at a common k (a tile row of a
  // lower x lower product): the row's workgroups then meet at the same k and share their A panel
  if constexpr (EDGE) {
    gemm_mainloop_regstage<R, A_KMAJOR, B_KMAJOR, T>(A, lda, B, ldb, M, N, row0, col0, kbeg, kend, smem, acc);
    return;
  } else {
    constexpr int MI = T / 32, WT = T / 2, KT = Real<R>::KT, LT = KT * T, NKK = KT / 4;
    constexpr int D = NS - 1;         // prefetch distance in tiles
    constexpr int OPS = 2 * MI;       // DMA instructions per wave per tile
    static_assert(NS >= 2 && (D - 1) * OPS < 64, "vmcnt is a 6-bit counter");
    using Img = LdsImage<R, T>;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fk = lane >> 4;
    row0 = __builtin_amdgcn_readfirstlane(row0);
    col0 = __builtin_amdgcn_readfirstlane(col0);
    kbeg = __builtin_amdgcn_readfirstlane(kbeg);
    kend = __builtin_amdgcn_readfirstlane(kend);
    if (kbeg >= kend) return;
    const int ntile = (kend - kbeg) / KT;

    TileDma<R, A_KMAJOR, T> da;
    TileDma<R, B_KMAJOR, T> db;
    da.init(A, lda, row0, kdesc ? kend - KT : kbeg, wave, lane);
    db.init(B, ldb, col0, kdesc ? kend - KT : kbeg, wave, lane);
    if (kdesc) {
      da.step = -da.step;
      db.step = -db.step;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;  // low 32 bits of a generic LDS pointer = LDS address
    constexpr uint32_t LTB = LT * sizeof(R);
    auto issue = [&](int stage) {
      const uint32_t img = lds0 + (uint32_t)stage * 2u * LTB;
      da.issue(img);
      db.issue(img + LTB);
      da.advance();
      db.advance();
    };

    __syncthreads();  // a previous tile's readers (stream-K) are done with every stage
#pragma unroll
    for (int t = 0; t < D; ++t)
      if (t < ntile) issue(t);
    if (ntile >= D) dma_wait_barrier<(D - 1) * OPS>();  // tile 0 of every wave has landed
    else dma_wait_barrier<0>();

    // MFMA fragments are double-buffered in registers across the kk sub-steps: the ds_reads of
    // sub-step kk+1 are issued BEFORE the 16 MFMAs of sub-step kk.  The single barrier of a K
    // step sits before the MFMAs of the last sub-step and is followed by the first fragment
    // reads of the next stage, so both waits are covered by those MFMAs; the DMA of tile s+D is
    // issued at the top of step s into the stage step s-1 read (its readers passed that barrier).
    R fa[2][MI], fb[2][MI];
    auto frag = [&](const R* S, int kk, R (&a)[MI], R (&b)[MI]) {
      const int k = kk * 4 + fk;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        a[i] = S[Img::template at<A_KMAJOR>(wm * WT + i * 16 + fr, k)];
        b[i] = S[LT + Img::template at<B_KMAJOR>(wn * WT + i * 16 + fr, k)];
      }
    };
    frag(smem, 0, fa[0], fb[0]);

    int cs = 0;  // stage holding tile s
    for (int s = 0; s < ntile; ++s) {
      const bool more = (s + 1) < ntile;
      const int nx = (cs + 1 == NS) ? 0 : cs + 1;
      const R* Sc = smem + cs * 2 * LT;
      const R* Sn = smem + nx * 2 * LT;
      if (s + D < ntile && !(ABL & 2)) issue((cs + D) % NS);
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk + 1 < NKK) {
          frag(Sc, kk + 1, fa[nxt], fb[nxt]);
        } else if (more) {
          if (!(ABL & 1)) {
            // tile s+1 must have landed; tiles s+2 .. s+D may stay in flight (fewer at the tail)
            if (s + D < ntile) dma_wait_barrier<(D - 1) * OPS>();
            else dma_wait_barrier<0>();
          }
          frag(Sn, 0, fa[nxt], fb[nxt]);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < MI; ++ni) acc[mi][ni] = Real<R>::mfma(fa[cur][mi], fb[cur][ni], acc[mi][ni]);
      }
      cs = nx;
    }
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
