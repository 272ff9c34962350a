// General MFMA GEMM for gfx950 (fp64 and fp32 instances) with per-tile triangular k ranges,
// lower-only output, batching and split-K.  See common.h for the argument contract.
#include "gemm_core.h"

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <set>
#include <utility>

namespace gpfit {

// NS = LDS stages of the main loop (gemm_core.h): 2 everywhere except the "deep" small-tile
// instances (4 at T = 64, 8 at T = 32) the launcher picks when a launch has at most two
// workgroups per CU, i.e. when nothing else hides the load latency.
// Tile selection, k range and main loop of one workgroup: on return acc holds op(A) op(B) of tile (ti, tj) at
// (row0, col0) and C points at this batch / split's output; false when the block has no tile (schedule padding).
template <typename R, bool A_KMAJOR, bool B_KMAJOR, bool EDGE, int T, int NS>
__device__ __forceinline__ bool gemm_tile_compute(const GemmArgsT<R>& p, int tiles_n, int ntiles, R* smem,
                                                  typename Real<R>::acc_t (&acc)[T / 32][T / 32], int& ti, int& tj,
                                                  int& row0, int& col0, R*& C) {
  constexpr int KT = Real<R>::KT;

  // heaviest tiles first: with triangular operands the k range depends on the tile position,
  // so the launcher asks for the walk that starts with the long ones (shorter tail):
  // bit 0 = walk backwards, bit 1 = column-major (dense output only).
  const int bid = (p.reverse & 1) ? (ntiles - 1 - (int)blockIdx.x) : (int)blockIdx.x;
  if (p.sched != nullptr) {
    // XCD-aware schedule: the table says which tile this block id computes (-1: padding entry)
    const int e = p.sched[blockIdx.x];
    if (e < 0) return false;
    ti = e >> 16;
    tj = e & 0xffff;
  } else if (p.out_lower && (p.reverse & 2) && T == TILE) {
    // column-major walk of the lower triangle: the tiles of a tile column share op(B)'s panel and,
    // where the k range depends on the column only, stay at the same k (lock step in L2)
    const int nt = tiles_n;
    int j = (int)((2.0 * nt + 1.0 - sqrt((2.0 * nt + 1.0) * (2.0 * nt + 1.0) - 8.0 * (double)bid)) * 0.5);
    while (j > 0 && (long)j * nt - (long)j * (j - 1) / 2 > bid) --j;
    while ((long)(j + 1) * nt - (long)(j + 1) * j / 2 <= bid) ++j;
    tj = j;
    ti = j + (bid - (j * nt - j * (j - 1) / 2));
  } else if (p.out_lower) {
    lower_tile(bid, TILE / T, ti, tj);
  } else if (p.reverse & 2) {
    const int tiles_m = ntiles / tiles_n;
    tj = bid / tiles_m;
    ti = bid % tiles_m;
  } else {
    ti = bid / tiles_n;
    tj = bid % tiles_n;
  }
  row0 = ti * T;
  col0 = tj * T;
  const int b = blockIdx.y, z = blockIdx.z;
  const R* A;
  const R* B;
  if (p.nptr > 0) {
    A = p.Ap[b];
    B = p.Bp[b];
    C = p.Cp[b];
  } else {
    A = p.A + (int64_t)b * p.sA;
    B = p.B + (int64_t)b * p.sB;
    C = p.C + (int64_t)b * p.sC;
  }

  int kbeg = 0, kend = p.K;
  if (p.a_tri == 1) kend = min(kend, row0 + T);
  if (p.a_tri == 2) kbeg = max(kbeg, row0);
  if (p.b_tri == 1) kbeg = max(kbeg, col0);
  if (p.b_tri == 2) kend = min(kend, col0 + T);
  // tri bounds are multiples of T >= 32 = the largest K step, so they stay K-step aligned
  if (p.split_k > 1) {
    const int steps = max(0, kend - kbeg) / KT;
    const int per = (steps + p.split_k - 1) / p.split_k;
    const int s0 = min(steps, z * per), s1 = min(steps, (z + 1) * per);
    kend = kbeg + s1 * KT;
    kbeg = kbeg + s0 * KT;
    C += (int64_t)z * p.sC;
  }

#pragma unroll
  for (int i = 0; i < T / 32; ++i)
#pragma unroll
    for (int j = 0; j < T / 32; ++j) acc[i][j] = acc_zero<R>();

  // walk bit 2: k downwards (EDGE instances ignore it)
  gemm_mainloop<R, A_KMAJOR, B_KMAJOR, EDGE, T, 0, NS>(A, p.lda, B, p.ldb, p.M, p.N, row0, col0, kbeg, kend, smem, acc,
                                                       (p.reverse & 4) != 0);
  return true;
}

// EPI: fused epilogue compiled into this instance (common.h GemmArgsT::epi; 0 = the plain alpha / beta store).
template <typename R, bool A_KMAJOR, bool B_KMAJOR, bool EDGE, int T, int NS, int EPI = 0>
__device__ __forceinline__ void gemm_tile_body(const GemmArgsT<R>& p, int tiles_n, int ntiles, R* smem) {
  typename Real<R>::acc_t acc[T / 32][T / 32];
  int ti, tj, row0, col0;
  R* C;
  if (!gemm_tile_compute<R, A_KMAJOR, B_KMAJOR, EDGE, T, NS>(p, tiles_n, ntiles, smem, acc, ti, tj, row0, col0, C)) return;

  const R alpha = (R)p.alpha, beta = (p.split_k > 1) ? (R)0 : (R)p.beta;
  const int64_t ldc = p.ldc;
  const int M = p.M, N = p.N;
  if constexpr (EPI != 0) {
    // fused epilogues (common.h: 1 mirror, 2 tile norms, 4 dual update); the launcher has checked that the
    // launch is data-parallel on full tiles
    constexpr int epi = EPI;
    R* __restrict__ D = p.nptr > 0 ? p.auxp[blockIdx.y] : p.aux;
    double* __restrict__ sumsq = p.nptr > 0 ? p.sumsqp[blockIdx.y] : p.sumsq;
    double ss = 0.0;
    for_each_acc<R, T>(acc, row0, col0, [&](int row, int col, R v) {
      if (EDGE && !(row < M && col < N)) return;
      R o = alpha * v;
      R* c = C + (int64_t)row * ldc + col;
      if (epi & 4) {
        R* dd = D + (int64_t)row * ldc + col;
        const R d0 = *dd;
        o += d0;
        *dd = o + d0;
      } else if (beta != (R)0) {
        o += beta * (*c);
      }
      if (epi & 1) {
        // the diagonal tile's strict upper part comes from the transposed store of its lower part
        if (row >= col) *c = o;
        if (row > col) C[(int64_t)col * ldc + row] = o;
      } else {
        *c = o;
      }
      if (epi & 2) ss += (double)o * (double)o;
    });
    if (epi & 2) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ss += __shfl_down(ss, o);
      double* red = reinterpret_cast<double*>(smem);
      __syncthreads();  // every wave is done with the operand stages
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
      __syncthreads();
      if (threadIdx.x == 0) sumsq[(int64_t)ti * (ti + 1) / 2 + tj] = (red[0] + red[1]) + (red[2] + red[3]);
    }
    return;
  }
  if (beta == (R)0) {
    for_each_acc<R, T>(acc, row0, col0, [&](int row, int col, R v) {
      if (!EDGE || (row < M && col < N)) C[(int64_t)row * ldc + col] = alpha * v;
    });
  } else {
    for_each_acc<R, T>(acc, row0, col0, [&](int row, int col, R v) {
      if (!EDGE || (row < M && col < N)) {
        R* c = C + (int64_t)row * ldc + col;
        *c = alpha * v + beta * (*c);
      }
    });
  }
}

template <typename R, bool A_KMAJOR, bool B_KMAJOR, bool EDGE, int T, int NS = 2>
__global__ __launch_bounds__(GEMM_THREADS, (T == 128 || NS > 2 ? 2 : 4)) void gemm_mfma_kernel(GemmArgsT<R> p,
                                                                                               int tiles_n,
                                                                                               int ntiles) {
  __shared__ __attribute__((aligned(16))) R smem[2 * NS * Real<R>::KT * T];
  gemm_tile_body<R, A_KMAJOR, B_KMAJOR, EDGE, T, NS>(p, tiles_n, ntiles, smem);
}

// The same 128-tile body under its own name for launches that follow an XCD-aware schedule table
// (gemm_sched.hip): in the fit these are exactly T = L^-1 L_V (<R, false, true>) and Q = I - T T^T
// (<R, false, false>), one launch each per evaluation, so a profiler's per-kernel row for this name
// IS that launch.
#ifdef GPFIT_CLOCK_STAMPS
// Diagnostic build only (scripts/scratch/dev_gemm_clock.sh): shader cycles (s_memtime) and 100 MHz ticks
// (s_memrealtime) each workgroup of the last scheduled launch spent, [B_KMAJOR][block][2]; read back with
// hipMemcpyFromSymbol by gpfit_dev_gemm_clock.  No output value depends on them.
__device__ long long g_gemm_clock[2][4096][2];
#endif

template <typename R, bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_xcd_kernel(GemmArgsT<R> p, int tiles_n, int ntiles) {
  __shared__ __attribute__((aligned(16))) R smem[4 * Real<R>::KT * TILE];
#ifdef GPFIT_CLOCK_STAMPS
  const long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
#endif
  gemm_tile_body<R, A_KMAJOR, B_KMAJOR, false, TILE, 2>(p, tiles_n, ntiles, smem);
#ifdef GPFIT_CLOCK_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 4096) {
    g_gemm_clock[B_KMAJOR ? 1 : 0][blockIdx.x][0] = (long long)__builtin_amdgcn_s_memtime() - t0;
    g_gemm_clock[B_KMAJOR ? 1 : 0][blockIdx.x][1] = (long long)__builtin_amdgcn_s_memrealtime() - w0;
  }
#endif
}

// The 128-tile body with a fused epilogue (common.h GemmArgsT::epi), with or without a schedule table.
template <typename R, bool A_KMAJOR, bool B_KMAJOR, int EPI>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_epi_kernel(GemmArgsT<R> p, int tiles_n, int ntiles) {
  __shared__ __attribute__((aligned(16))) R smem[4 * Real<R>::KT * TILE];
#ifdef GPFIT_CLOCK_STAMPS
  const long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
#endif
  gemm_tile_body<R, A_KMAJOR, B_KMAJOR, false, TILE, 2, EPI>(p, tiles_n, ntiles, smem);
#ifdef GPFIT_CLOCK_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 4096 && EPI != 4) {
    g_gemm_clock[B_KMAJOR ? 1 : 0][blockIdx.x][0] = (long long)__builtin_amdgcn_s_memtime() - t0;
    g_gemm_clock[B_KMAJOR ? 1 : 0][blockIdx.x][1] = (long long)__builtin_amdgcn_s_memrealtime() - w0;
  }
#endif
}

// Tile size: 128 when that already gives the chip >= 1.5 waves of blocks, otherwise 64 / 32 so
// the small panels near the leaves of the recursion are not serialised on a handful of CUs.
template <typename R>
int gemm_pick_tile(const GemmArgsT<R>& a) {
  if (a.tile == 128 || a.tile == 64 || a.tile == 32) return a.tile;
  auto ntiles = [&](int T) {
    const long tm = (a.M + T - 1) / T, tn = (a.N + T - 1) / T;
    const long nb = (a.M + TILE - 1) / TILE;
    return (a.out_lower ? (long)lower_tile_count((int)nb, TILE / T) : tm * tn) * (a.nptr > 0 ? a.nptr : (a.batch > 0 ? a.batch : 1)) *
           (a.split_k > 1 ? a.split_k : 1);
  };
  static const long t128_min = getenv("GPFIT_T128_MIN") ? atol(getenv("GPFIT_T128_MIN")) : 384;
  // pointer batches with triangular operands: the tiles' k ranges differ by up to the matrix size and a batch has
  // no balanced (stream-K / table) schedule, so the 128-tile only pays once the launch runs for several rounds of
  // the chip (measured at 2048-sized blocks, executed TF/s: two problems 37-39 on 128-tiles against 54 on
  // 64-tiles launched one by one; four problems 55)
  static const long t128_tri_min = getenv("GPFIT_T128_TRI_MIN") ? atol(getenv("GPFIT_T128_TRI_MIN")) : 1024;
  const bool batch_tri = a.nptr > 0 && (a.a_tri || a.b_tri);
  if (ntiles(128) >= (batch_tri ? t128_tri_min : t128_min)) return 128;
  static const long t64_min = getenv("GPFIT_T64_MIN") ? atol(getenv("GPFIT_T64_MIN")) : 256;
  static const long t32_dim = getenv("GPFIT_T32_DIM") ? atol(getenv("GPFIT_T32_DIM")) : 1024;
  if (ntiles(64) >= t64_min) return 64;
  return (a.M <= t32_dim && a.N <= t32_dim) ? 32 : 64;
}

constexpr int HALF_OCC_LDS = 17 * 1024;
// a kernel whose static + dynamic LDS exceeds 64 KiB needs the limit raised once per (function, device)
static void allow_dynamic_lds(const void* fn) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int device = 0;
  (void)hipGetDevice(&device);
  std::lock_guard<std::mutex> lock(mu);
  if (done.insert({fn, device}).second)
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, HALF_OCC_LDS);
}

template <typename R, int T>
static void launch_T(const GemmArgsT<R>& p, hipStream_t s) {
  const int tm = (p.M + T - 1) / T, tn = (p.N + T - 1) / T;
  const int tiles = p.out_lower ? lower_tile_count((p.M + TILE - 1) / TILE, TILE / T) : tm * tn;
  const bool edge = (p.M % T) || (p.N % T) || (p.out_lower && (p.M % TILE));
  dim3 grid(p.sched ? p.sched_blocks : (p.tile_limit > 0 ? std::min(p.tile_limit, tiles) : tiles), p.batch,
            p.split_k > 1 ? p.split_k : 1);
  dim3 block(GEMM_THREADS);
  constexpr int DEEP = (T == 128) ? 2 : (T == 64 ? 4 : 8);
  static const int deep_max = getenv("GPFIT_DEEP_MAX") ? atoi(getenv("GPFIT_DEEP_MAX")) : 512;  // tuning knob
  const bool deep = DEEP > 2 && !edge && (long)grid.x * grid.y * grid.z <= deep_max;
  // half-occupancy launches (T = 128 only): 64 KiB static + 17 KiB of unused dynamic LDS = 81 KiB > 160 / 2
  const bool half = (T == TILE) && p.half_occ && !deep;
#define GP_LAUNCH(AK, BK, ED)                                                                          \
  do {                                                                                                 \
    if (deep) hipLaunchKernelGGL((gemm_mfma_kernel<R, AK, BK, ED, T, (ED ? 2 : DEEP)>), grid, block, 0, s, p, tn, tiles); \
    else if (half) {                                                                                   \
      allow_dynamic_lds(reinterpret_cast<const void*>(gemm_mfma_kernel<R, AK, BK, ED, T, 2>));          \
      hipLaunchKernelGGL((gemm_mfma_kernel<R, AK, BK, ED, T, 2>), grid, block, HALF_OCC_LDS, s, p, tn, tiles); \
    } else hipLaunchKernelGGL((gemm_mfma_kernel<R, AK, BK, ED, T, 2>), grid, block, 0, s, p, tn, tiles); \
  } while (0)
  if (p.epi) {
    // fused epilogues exist for the layouts the fit uses: T = L^-1 L_V with tile norms, Q = -T T^T mirrored,
    // H = Q21 A + Z21 with the dual update (gemm_epilogue_ok has checked the combination)
    if constexpr (T == TILE) {
      const int key = p.epi * 4 + (p.a_kmajor ? 2 : 0) + (p.b_kmajor ? 1 : 0);
      if (key == 2 * 4 + 1) hipLaunchKernelGGL((gemm_epi_kernel<R, false, true, 2>), grid, block, 0, s, p, tn, tiles);
      else if (key == 1 * 4 + 0) hipLaunchKernelGGL((gemm_epi_kernel<R, false, false, 1>), grid, block, 0, s, p, tn, tiles);
      else if (key == 4 * 4 + 1) hipLaunchKernelGGL((gemm_epi_kernel<R, false, true, 4>), grid, block, 0, s, p, tn, tiles);
    }
    return;
  }
  if (p.sched && T == TILE && !edge) {
    switch ((p.a_kmajor ? 2 : 0) | (p.b_kmajor ? 1 : 0)) {
      case 0: hipLaunchKernelGGL((gemm_xcd_kernel<R, false, false>), grid, block, 0, s, p, tn, tiles); break;
      case 1: hipLaunchKernelGGL((gemm_xcd_kernel<R, false, true>), grid, block, 0, s, p, tn, tiles); break;
      case 2: hipLaunchKernelGGL((gemm_xcd_kernel<R, true, false>), grid, block, 0, s, p, tn, tiles); break;
      default: hipLaunchKernelGGL((gemm_xcd_kernel<R, true, true>), grid, block, 0, s, p, tn, tiles); break;
    }
    return;
  }
  const int sel = (p.a_kmajor ? 4 : 0) | (p.b_kmajor ? 2 : 0) | (edge ? 1 : 0);
  switch (sel) {
    case 0: GP_LAUNCH(false, false, false); break;
    case 1: GP_LAUNCH(false, false, true); break;
    case 2: GP_LAUNCH(false, true, false); break;
    case 3: GP_LAUNCH(false, true, true); break;
    case 4: GP_LAUNCH(true, false, false); break;
    case 5: GP_LAUNCH(true, false, true); break;
    case 6: GP_LAUNCH(true, true, false); break;
    case 7: GP_LAUNCH(true, true, true); break;
  }
#undef GP_LAUNCH
}

template <typename R>
bool gemm_epilogue_ok(const GemmArgsT<R>& a) {
  if (a.split_k > 1 || (a.batch > 1 && a.nptr <= 0) || a.M <= 0 || a.N <= 0) return false;
  const int T = gemm_pick_tile(a);
  if ((a.M % T) || (a.N % T)) return false;                                    // full tiles only
  if ((a.epi & 1) && (!a.out_lower || a.M != a.N)) return false;
  if ((a.epi & 2) && (T != TILE || !a.out_lower)) return false;
  if ((a.epi & 4) && a.nptr <= 0 && a.aux == nullptr) return false;
  // instances that exist (launch_T): 128-tile, row-major A, and per mode the operand layout the fit uses
  if (T != TILE || a.a_kmajor || a.half_occ) return false;
  if (!((a.epi == 2 && a.b_kmajor) || (a.epi == 1 && !a.b_kmajor) || (a.epi == 4 && a.b_kmajor))) return false;
  if (a.nptr > 0) return true;   // pointer batches are always data-parallel
  if (T == TILE && a.tile_limit == 0) {
    if ((a.reverse & 8) && gemm_xcd_applies(a)) return true;
    if (gemm_streamk_applies(a)) return gemm_streamk_carries(a);
  }
  return true;
}

template <typename R>
int gemm_sumsq_entries(const GemmArgsT<R>& a) {
  if (!(a.epi & 2) || !gemm_epilogue_ok(a)) return 0;
  const int t = a.M / TILE, nt = t * (t + 1) / 2;
  if (a.nptr > 0 || a.half_occ || a.tile_limit != 0 || a.batch > 1) return nt;
  if ((a.reverse & 8) && gemm_xcd_applies(a)) return nt;
  return gemm_streamk_applies(a) ? 33 * nt : nt;
}
template int gemm_sumsq_entries<double>(const GemmArgsT<double>&);
template int gemm_sumsq_entries<float>(const GemmArgsT<float>&);
template bool gemm_epilogue_ok<double>(const GemmArgsT<double>&);
template bool gemm_epilogue_ok<float>(const GemmArgsT<float>&);

template <typename R>
int launch_gemm(const GemmArgsT<R>& a, hipStream_t s) {
  if (a.M <= 0 || a.N <= 0) return 0;
  if (a.epi && !gemm_epilogue_ok(a)) {
    set_error("launch_gemm: this launch cannot carry a fused epilogue (ask gemm_epilogue_ok first)");
    return -3;
  }
  if (a.nptr > 0) return launch_gemm_plain(a, s);   // pointer batches are data-parallel launches
  if (a.half_occ) return launch_gemm_plain(a, s);
  if (a.tile_limit == 0 && gemm_pick_tile(a) == TILE && a.batch <= 1) {
    if (a.reverse & 8) {
      const int rc = launch_gemm_xcd(a, s);    // XCD-aware data-parallel schedule (gemm_sched.hip)
      if (rc <= 0) return rc;
    }
    const int rc = launch_gemm_streamk(a, s);  // large launches: balanced schedules (gemm_streamk.hip)
    if (rc <= 0) return rc;
  }
  return launch_gemm_plain(a, s);
}

template <typename R>
int launch_gemm_plain(const GemmArgsT<R>& a, hipStream_t s) {
  if (a.M <= 0 || a.N <= 0) return 0;
  // odd M/N are fine for the stores; k-major operands are then read up to one 16-byte chunk past
  // M/N, which internal callers cover with zero padding (the public gpfit_dgemm insists on even).
  constexpr int EPC = 16 / (int)sizeof(R);
  if (a.K % ktile_of<R>() != 0 || (a.lda % EPC) || (a.ldb % EPC)) {
    set_error("launch_gemm: K must be a multiple of the K step and lda, ldb multiples of 16 bytes");
    return -3;
  }
  if (a.out_lower && a.M != a.N) {
    set_error("launch_gemm: out_lower needs a square output");
    return -3;
  }
  GemmArgsT<R> p = a;
  if (p.nptr > 0) {
    if (p.nptr > GEMM_MAXB || p.sched || p.split_k > 1) {
      set_error("launch_gemm: a pointer batch holds at most GEMM_MAXB plain problems");
      return -3;
    }
    p.batch = p.nptr;
  }
  if (p.batch <= 0) p.batch = 1;
  switch (gemm_pick_tile(p)) {
    case 128: launch_T<R, 128>(p, s); break;
    case 64: launch_T<R, 64>(p, s); break;
    default: launch_T<R, 32>(p, s); break;
  }
  GP_HIP(hipGetLastError());
  return 0;
}

// ---- two structurally different small-tile pointer batches in one launch (common.h: launch_gemm_pair)
template <typename R>
struct GemmPairT {
  GemmArgsT<R> g[2];
  int tiles[2], tiles_n[2];
};

template <typename R, int T, int NS>
__global__ __launch_bounds__(GEMM_THREADS, (NS > 2 ? 2 : 4)) void gemm_pair_kernel(GemmPairT<R> q) {
  __shared__ __attribute__((aligned(16))) R smem[2 * NS * Real<R>::KT * T];
  const int w = blockIdx.z;
  const GemmArgsT<R>& p = q.g[w];
  if ((int)blockIdx.x >= q.tiles[w] || (int)blockIdx.y >= p.batch) return;
  if (p.b_kmajor) gemm_tile_body<R, false, true, false, T, NS>(p, q.tiles_n[w], q.tiles[w], smem);
  else gemm_tile_body<R, false, false, false, T, NS>(p, q.tiles_n[w], q.tiles[w], smem);
}

// the block tile both members would take on their own (32 or 64), 0: not a member of a pair
template <typename R>
static int pair_member_tile(const GemmArgsT<R>& a) {
  constexpr int EPC = 16 / (int)sizeof(R);
  if (a.nptr <= 0 || a.nptr > GEMM_MAXB || a.epi || a.split_k > 1 || a.sched || a.half_occ || a.tile_limit || a.a_kmajor) return 0;
  if (a.M <= 0 || a.N <= 0 || (a.M % 64) || (a.N % 64) || (a.out_lower && ((a.M % TILE) || a.M != a.N))) return 0;
  if (a.K % ktile_of<R>() != 0 || (a.lda % EPC) || (a.ldb % EPC)) return 0;
  const int T = gemm_pick_tile(a);
  return (T == 32 || T == 64) ? T : 0;
}

template <typename R>
bool gemm_pair_ok(const GemmArgsT<R>& a, const GemmArgsT<R>& b) {
  static const bool off = getenv("GPFIT_NO_PAIR") != nullptr;   // tuning knob: the two launches on their own
  static const bool no64 = getenv("GPFIT_NO_PAIR64") != nullptr;
  if (off) return false;
  const int ta = pair_member_tile(a), tb = pair_member_tile(b);
  return ta != 0 && ta == tb && !(no64 && ta == 64);
}
template bool gemm_pair_ok<double>(const GemmArgsT<double>&, const GemmArgsT<double>&);
template bool gemm_pair_ok<float>(const GemmArgsT<float>&, const GemmArgsT<float>&);

template <typename R>
int launch_gemm_pair(const GemmArgsT<R>& a, const GemmArgsT<R>& b, hipStream_t s) {
  if (!gemm_pair_ok(a, b)) {
    set_error("launch_gemm_pair: the two problems cannot share a launch (ask gemm_pair_ok first)");
    return -3;
  }
  static_assert(sizeof(GemmPairT<R>) <= 4096, "kernel arguments are limited to 4 KiB");
  const int T = pair_member_tile(a);
  GemmPairT<R> q;
  int gx = 0, gy = 0;
  long total = 0;
  for (int w = 0; w < 2; ++w) {
    q.g[w] = w == 0 ? a : b;
    GemmArgsT<R>& p = q.g[w];
    p.batch = p.nptr;
    const int tm = p.M / T, tn = p.N / T;
    q.tiles_n[w] = tn;
    q.tiles[w] = p.out_lower ? lower_tile_count(p.M / TILE, TILE / T) : tm * tn;
    gx = std::max(gx, q.tiles[w]);
    gy = std::max(gy, p.batch);
    total += (long)q.tiles[w] * p.batch;
  }
  const dim3 grid(gx, gy, 2), block(GEMM_THREADS);
  // stages as launch_T picks them: the deep pipeline when the launch has at most two workgroups per CU
  static const int deep_max = getenv("GPFIT_DEEP_MAX") ? atoi(getenv("GPFIT_DEEP_MAX")) : 512;
  if (T == 32 && total <= deep_max) hipLaunchKernelGGL((gemm_pair_kernel<R, 32, 8>), grid, block, 0, s, q);
  else if (T == 32) hipLaunchKernelGGL((gemm_pair_kernel<R, 32, 2>), grid, block, 0, s, q);
  else if (total <= deep_max) hipLaunchKernelGGL((gemm_pair_kernel<R, 64, 4>), grid, block, 0, s, q);
  else hipLaunchKernelGGL((gemm_pair_kernel<R, 64, 2>), grid, block, 0, s, q);
  GP_HIP(hipGetLastError());
  return 0;
}
template int launch_gemm_pair<double>(const GemmArgsT<double>&, const GemmArgsT<double>&, hipStream_t);
template int launch_gemm_pair<float>(const GemmArgsT<float>&, const GemmArgsT<float>&, hipStream_t);

template int launch_gemm<double>(const GemmArgsT<double>&, hipStream_t);
template int launch_gemm<float>(const GemmArgsT<float>&, hipStream_t);
template int launch_gemm_plain<double>(const GemmArgsT<double>&, hipStream_t);
template int launch_gemm_plain<float>(const GemmArgsT<float>&, hipStream_t);
template int gemm_pick_tile<double>(const GemmArgsT<double>&);
template int gemm_pick_tile<float>(const GemmArgsT<float>&);

}  // namespace gpfit

#ifdef GPFIT_CLOCK_STAMPS
extern "C" int gpfit_dev_gemm_clock(long long* host_out /* [2][4096][2] */) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(gpfit::g_gemm_clock), sizeof(long long) * 2 * 4096 * 2) == hipSuccess ? 0 : -1;
}
#endif
