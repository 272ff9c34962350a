// Stream-K schedule for the large fp64 MFMA GEMM launches.
//
// The data-parallel launch (gemm.hip) gives every 128 x 128 output tile to one workgroup.  With
// 2 workgroups per CU the chip runs 512 tiles at a time, so a 2080-tile SYRK takes 5 rounds
// for 4.06 rounds of work, a 528-tile one 2 rounds for 1.03, and triangular operands (k range
// proportional to the tile position) leave most CUs idle behind a few long tiles (with the
// register-staged main loop of the time: 51-58 TF/s against 66 for the dense 4096-tile case).
//
// Here the (tile, k-step) iteration space is flattened in the launch's tile-walk order and cut
// into one equal contiguous share per resident workgroup.  A workgroup whose share starts or
// ends inside a tile writes that tile's partial accumulator to a workspace slot (at most two
// per workgroup); a second tiny kernel adds the partials of each split tile IN A FIXED ORDER
// and applies alpha/beta, so results are bit-reproducible (no atomics).  The plan (tile table,
// fix-up lists) depends only on the launch shape and is cached on the device.
#include "gemm_core.h"

#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace gpfit {

struct SkTile {
  int row0, col0, kbeg, ksteps;
  int prefix;  // k-steps of all tiles before this one in walk order (< 2^31: 4096 tiles x 512 steps)
};

struct SkPlan {
  SkTile* tiles = nullptr;   // device
  int ntiles = 0;
  int total = 0;
  int blocks = 0;
  int per_block = 0;
  int* fix_tile = nullptr;   // device: split tile ids
  int* fix_ptr = nullptr;    // device: CSR offsets into fix_slot
  int* fix_slot = nullptr;   // device: workspace slots in accumulation order
  int nfix = 0;
};

template <typename R>
struct SkParams {
  const R* A;
  const R* B;
  R* C;
  int64_t lda, ldb, ldc;
  int M, N;
  double alpha, beta;
  const SkTile* tiles;
  int ntiles;
  int total, per_block;
  R* partial;
  // fused epilogue (common.h GemmArgsT::epi, bit 2 only; lower square output, every tile on the stream-K
  // schedule).  Tile norms: a tile finished by one workgroup leaves its sum of squares in sumsq[idx], idx =
  // ti (ti + 1) / 2 + tj; a split tile is finished by the 32 bands of the fix-up kernel, which leave theirs in
  // sumsq[nt_all + 32 idx + band].  The other entries of a tile are written as zero, so the sum over all
  // 33 nt_all entries is the squared Frobenius norm whatever the cut.
  double* sumsq;
  int nt_all;
};

constexpr int SK_SLOTS = 512;  // resident workgroups: 256 CUs x 2 (189 VGPRs, 64 KiB LDS each)

// one segment of a tile's k range (its own function so that the kernel's epilogue variants share ONE call site of the
// main loop: the host pass of the compiler rejects a second instantiation context of the same main-loop specialisation)
template <typename R, bool A_KMAJOR, bool B_KMAJOR>
__device__ __forceinline__ void sk_segment(const SkParams<R>& p, const SkTile& tl, int kb, int ke, R* smem,
                                           typename Real<R>::acc_t (&acc)[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = acc_zero<R>();
  gemm_mainloop<R, A_KMAJOR, B_KMAJOR, false, TILE>(p.A, p.lda, p.B, p.ldb, p.M, p.N, tl.row0, tl.col0, kb, ke, smem, acc);
}

template <typename R, bool A_KMAJOR, bool B_KMAJOR, int EPI = 0>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_streamk_kernel(SkParams<R> p) {
  constexpr int KT = Real<R>::KT;
  __shared__ __attribute__((aligned(16))) R smem[4 * KT * TILE];
  int it = (int)blockIdx.x * p.per_block;
  const int it_end = min(p.total, it + p.per_block);
  if (it >= it_end) return;
  // first tile whose range contains `it` (binary search on the prefix sums)
  int lo = 0, hi = p.ntiles - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (p.tiles[mid].prefix <= it) lo = mid; else hi = mid - 1;
  }
  int t = lo, nseg = 0;
  while (it < it_end) {
    const SkTile tl = p.tiles[t];
    const int tbeg = tl.prefix, tend = tbeg + tl.ksteps;
    const int s1 = min(it_end, tend);
    const int kb = tl.kbeg + (it - tbeg) * KT, ke = tl.kbeg + (s1 - tbeg) * KT;
    typename Real<R>::acc_t acc[4][4];
    sk_segment<R, A_KMAJOR, B_KMAJOR>(p, tl, kb, ke, smem, acc);
    bool whole_done = false;
    if constexpr (EPI != 0) {
     if (it == tbeg && s1 == tend) {
      whole_done = true;
      R* __restrict__ C = p.C;
      const int64_t ldc = p.ldc;
      const R alpha = (R)p.alpha, beta = (R)p.beta;
      double ss = 0.0;
      for_each_acc<R, TILE>(acc, tl.row0, tl.col0, [&](int row, int col, R v) {
        R* c = C + (int64_t)row * ldc + col;
        R o = alpha * v;
        if (beta != (R)0) o += beta * (*c);
        *c = o;
        ss += (double)o * (double)o;
      });
      if (EPI & 2) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_down(ss, o);
        double* red = reinterpret_cast<double*>(smem);
        __syncthreads();  // every wave is done with the operand stages
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
        __syncthreads();
        const int ti = tl.row0 / TILE, tj = tl.col0 / TILE;
        const int64_t idx = (int64_t)ti * (ti + 1) / 2 + tj;
        if (threadIdx.x == 0) p.sumsq[idx] = (red[0] + red[1]) + (red[2] + red[3]);
        if (threadIdx.x < 32) p.sumsq[p.nt_all + 32 * idx + threadIdx.x] = 0.0;
        __syncthreads();  // red lives in the first operand stage
      }
     }
    }
    if (whole_done) {
    } else if (it == tbeg && s1 == tend) {
      R* __restrict__ C = p.C;
      const int64_t ldc = p.ldc;
      const R alpha = (R)p.alpha, beta = (R)p.beta;
      if (beta == (R)0) {
        for_each_acc<R, TILE>(acc, tl.row0, tl.col0,
                              [&](int row, int col, R v) { C[(int64_t)row * ldc + col] = alpha * v; });
      } else {
        for_each_acc<R, TILE>(acc, tl.row0, tl.col0, [&](int row, int col, R v) {
          R* c = C + (int64_t)row * ldc + col;
          *c = alpha * v + beta * (*c);
        });
      }
    } else {
      R* __restrict__ P = p.partial + ((int64_t)2 * blockIdx.x + (nseg > 0 ? 1 : 0)) * (TILE * TILE);
      for_each_acc<R, TILE>(acc, 0, 0, [&](int row, int col, R v) { P[row * TILE + col] = v; });
    }
    ++nseg;
    it = s1;
    ++t;
  }
}

// C tile = alpha * (sum of its partial slots, in plan order) + beta * C
template <typename R, int EPI = 0>
__global__ __launch_bounds__(256) void streamk_fixup_kernel(SkParams<R> p, const int* __restrict__ fix_tile,
                                                            const int* __restrict__ fix_ptr,
                                                            const int* __restrict__ fix_slot) {
  // grid = (split tiles, 32): each workgroup sums a 4-row band (two elements per thread as one 16-byte access
  // in fp64), so that a launch with few split tiles but many partials per tile (short tails) still spreads over
  // the chip; the partials of an element are loaded four at a time and added in plan order (the latency of
  // the loads overlaps, the order of the additions -- hence the bits of the result -- does not change)
  using V = typename Real<R>::vec_t;
  constexpr int EPC = Real<R>::EPC;
  const SkTile tl = p.tiles[fix_tile[blockIdx.x]];
  const int s0 = fix_ptr[blockIdx.x], s1 = fix_ptr[blockIdx.x + 1];
  const R alpha = (R)p.alpha, beta = (R)p.beta;
  const int band = TILE / (int)gridDim.y;
  double ss = 0.0;
  for (int e = threadIdx.x * EPC; e < band * TILE; e += blockDim.x * EPC) {
    const int r = blockIdx.y * band + (e >> 7), c = e & 127;
    const int64_t off = (int64_t)r * TILE + c;
    V sum = V{};
    int s = s0;
    for (; s + 4 <= s1; s += 4) {
      const V v0 = *reinterpret_cast<const V*>(p.partial + (int64_t)fix_slot[s] * (TILE * TILE) + off);
      const V v1 = *reinterpret_cast<const V*>(p.partial + (int64_t)fix_slot[s + 1] * (TILE * TILE) + off);
      const V v2 = *reinterpret_cast<const V*>(p.partial + (int64_t)fix_slot[s + 2] * (TILE * TILE) + off);
      const V v3 = *reinterpret_cast<const V*>(p.partial + (int64_t)fix_slot[s + 3] * (TILE * TILE) + off);
      sum += v0; sum += v1; sum += v2; sum += v3;
    }
    for (; s < s1; ++s) sum += *reinterpret_cast<const V*>(p.partial + (int64_t)fix_slot[s] * (TILE * TILE) + off);
    V* cp = reinterpret_cast<V*>(p.C + (int64_t)(tl.row0 + r) * p.ldc + tl.col0 + c);
    const V val = (beta == (R)0) ? alpha * sum : alpha * sum + beta * (*cp);
    *cp = val;
    if constexpr ((EPI & 2) != 0) {
#pragma unroll
      for (int q = 0; q < EPC; ++q) ss += (double)val[q] * (double)val[q];
    }
  }
  if constexpr ((EPI & 2) != 0) {
    __shared__ double red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_down(ss, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int ti = tl.row0 / TILE, tj = tl.col0 / TILE;
      const int64_t idx = (int64_t)ti * (ti + 1) / 2 + tj;
      p.sumsq[p.nt_all + 32 * idx + blockIdx.y] = (red[0] + red[1]) + (red[2] + red[3]);
      if (blockIdx.y == 0) p.sumsq[idx] = 0.0;
    }
  }
}

// Plans hold device pointers, so the key carries the device: (device, M, N, K, lower, a_tri, b_tri, walk, element size)
using SkKey = std::tuple<int, int, int, int, int, int, int, int, int>;
static std::map<SkKey, SkPlan> g_plans;
static std::mutex g_plan_mutex;
// Fallback partial-tile workspaces of the context-free gpfit_dgemm, one per (device, stream): two
// streams issuing large GEMMs never share partial tiles.  Sized for fp64, used by fp32 as well.
static std::map<std::pair<int, hipStream_t>, void*> g_workspace;

template <typename R>
static int build_plan(const GemmArgsT<R>& a, int first, SkPlan& plan) {
  constexpr int KT = 128 / (int)sizeof(R);
  const int tm = a.M / TILE, tn = a.N / TILE;
  const int all_tiles = a.out_lower ? tm * (tm + 1) / 2 : tm * tn;
  const int ntiles = all_tiles - first;
  std::vector<SkTile> tiles;
  tiles.reserve(ntiles);
  long long prefix = 0;
  for (int b = first; b < all_tiles; ++b) {
    const int bid = (a.reverse & 1) ? (all_tiles - 1 - b) : b;
    int ti, tj;
    if (a.out_lower) tri_tile(bid, ti, tj);
    else if (a.reverse & 2) { tj = bid / tm; ti = bid % tm; }
    else { ti = bid / tn; tj = bid % tn; }
    int kb = 0, ke = a.K;
    if (a.a_tri == 1) ke = std::min(ke, ti * TILE + TILE);
    if (a.a_tri == 2) kb = std::max(kb, ti * TILE);
    if (a.b_tri == 1) kb = std::max(kb, tj * TILE);
    if (a.b_tri == 2) ke = std::min(ke, tj * TILE + TILE);
    const int ks = std::max(0, ke - kb) / KT;
    if (ks == 0) return 1;  // empty tiles would need a beta-only pass: leave those launches to gemm.hip
    tiles.push_back(SkTile{ti * TILE, tj * TILE, kb, ks, (int)prefix});
    prefix += ks;
  }
  plan.ntiles = ntiles;
  if (prefix >= (1LL << 31)) return 1;
  plan.total = (int)prefix;
  // at most 16 shares per tile: finer cuts only add partial-tile traffic and fix-up work
  plan.blocks = (int)std::min<long long>(std::min<long long>(SK_SLOTS, (long long)ntiles * 16), prefix);
  plan.per_block = (int)((prefix + plan.blocks - 1) / plan.blocks);
  // replay the kernel's walk to list, per split tile, the slots in accumulation (block) order
  std::vector<std::vector<int>> slots(ntiles);
  int t = 0;
  for (int b = 0; b < plan.blocks; ++b) {
    int it = b * plan.per_block;
    const int it_end = std::min(plan.total, it + plan.per_block);
    if (it >= it_end) break;
    while (tiles[t].prefix + tiles[t].ksteps <= it) ++t;
    int tt = t, nseg = 0;
    while (it < it_end) {
      const int tbeg = tiles[tt].prefix, tend = tbeg + tiles[tt].ksteps;
      const int s1 = std::min(it_end, tend);
      if (!(it == tbeg && s1 == tend)) slots[tt].push_back(2 * b + (nseg > 0 ? 1 : 0));
      ++nseg;
      it = s1;
      ++tt;
    }
  }
  std::vector<int> fix_tile, fix_ptr{0}, fix_slot;
  for (int i = 0; i < ntiles; ++i)
    if (!slots[i].empty()) {
      fix_tile.push_back(i);
      fix_slot.insert(fix_slot.end(), slots[i].begin(), slots[i].end());
      fix_ptr.push_back((int)fix_slot.size());
    }
  plan.nfix = (int)fix_tile.size();
  GP_HIP(hipMalloc((void**)&plan.tiles, tiles.size() * sizeof(SkTile)));
  GP_HIP(hipMemcpy(plan.tiles, tiles.data(), tiles.size() * sizeof(SkTile), hipMemcpyHostToDevice));
  if (plan.nfix) {
    GP_HIP(hipMalloc((void**)&plan.fix_tile, fix_tile.size() * sizeof(int)));
    GP_HIP(hipMalloc((void**)&plan.fix_ptr, fix_ptr.size() * sizeof(int)));
    GP_HIP(hipMalloc((void**)&plan.fix_slot, fix_slot.size() * sizeof(int)));
    GP_HIP(hipMemcpy(plan.fix_tile, fix_tile.data(), fix_tile.size() * sizeof(int), hipMemcpyHostToDevice));
    GP_HIP(hipMemcpy(plan.fix_ptr, fix_ptr.data(), fix_ptr.size() * sizeof(int), hipMemcpyHostToDevice));
    GP_HIP(hipMemcpy(plan.fix_slot, fix_slot.data(), fix_slot.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  return 0;
}

// Returns 0 when the launch was issued here, 1 when the caller should use the plain
// data-parallel launch, < 0 on error.  Two uses:
//  * operands triangular on both sides (k range of a tile ~ distance from the diagonal): pure
//    stream-K over all tiles (+8 % on L^-1 L_V, T T^T, L^-T R);
//  * uniform k range whose tile count leaves a short last round (2080 = 4 x 512 + 32): the full
//    rounds stay data-parallel -- workgroups that start together walk k in lock step and share
//    operand panels in L2, which stream-K's staggered shares give up -- and only the tail tiles
//    are cut along k over the whole chip.
// first: number of leading tiles that stay data-parallel (tails of uniform launches); -1: not a stream-K launch
template <typename R>
static int streamk_first_tile(const GemmArgsT<R>& a) {
  static const bool disabled = getenv("GPFIT_NO_STREAMK") != nullptr;
  if (disabled) return -1;
  if ((a.M % TILE) || (a.N % TILE) || a.split_k > 1 || a.batch > 1 || a.nptr > 0 || (a.tile && a.tile != TILE)) return -1;
  const long tm = a.M / TILE, tn = a.N / TILE;
  const int ntiles = (int)(a.out_lower ? tm * (tm + 1) / 2 : tm * tn);
  static const int sk_min = getenv("GPFIT_SK_MIN_TILES") ? atoi(getenv("GPFIT_SK_MIN_TILES")) : 384;
  static const int sk_all = getenv("GPFIT_SK_ALL") ? 1 : 0;  // experiment: stream-K for every eligible launch
  if (ntiles < sk_min || (long)a.K < 1024) return -1;  // small launches: latency-, not balance-bound
  int first = 0;
  // classes of launches that take the stream-K schedule (tuning knob, bit mask): 1 operands
  // triangular on both sides, 2 lower output with an upper-triangular op(A), 4 tails of uniform
  // launches.  Class 2 is off by default: since the LDS-DMA main loop its data-parallel launch
  // (heavy rows first) is the faster one (2.86 vs 3.04 ms at N = 8192).
  static const int sk_classes = getenv("GPFIT_SK_CLASSES") ? atoi(getenv("GPFIT_SK_CLASSES")) : 5;
  const bool cls1 = (a.a_tri != 0 && a.b_tri != 0), cls2 = (a.out_lower && a.a_tri == 2 && a.b_tri == 0);
  if ((cls1 && !(sk_classes & 1)) || (cls2 && !(sk_classes & 2))) return -1;
  const bool both_tri = cls1 || cls2;
  if (!both_tri && !(sk_classes & 4)) return -1;
  if (!both_tri && !(sk_all && ntiles < SK_SLOTS)) {
    if (a.a_tri || a.b_tri) return -1;       // one-sided triangles: the heavy-first walk already balances
    const int tail = ntiles % SK_SLOTS;
    if (tail == 0 || tail >= 384 || ntiles < SK_SLOTS) return -1;
    first = ntiles - tail;
  }
  return first;
}

template <typename R>
bool gemm_streamk_applies(const GemmArgsT<R>& a) { return streamk_first_tile(a) >= 0; }
template bool gemm_streamk_applies<double>(const GemmArgsT<double>&);
template bool gemm_streamk_applies<float>(const GemmArgsT<float>&);

// Fused epilogue on the stream-K schedule: the tile norms (2) of a square lower output whose tiles ALL take the
// stream-K schedule (T = L^-1 L_V of a unit below the size where the XCD-aware tables take over).  The mirrored
// store (1) was built for this schedule too and taken out again: the transposed stores of an accumulator tile are
// 32-byte fragments, which cost Q's launch 50 us and its fix-up 16 at N = 4096 against 28 us for the separate
// symmetrisation pass (which transposes through LDS).
template <typename R>
bool gemm_streamk_carries(const GemmArgsT<R>& a) {
  if (streamk_first_tile(a) != 0) return false;
  if (!a.out_lower || a.M != a.N || a.a_kmajor) return false;
  return a.epi == 2 && a.b_kmajor && a.sumsq != nullptr;
}
template bool gemm_streamk_carries<double>(const GemmArgsT<double>&);
template bool gemm_streamk_carries<float>(const GemmArgsT<float>&);

template <typename R>
int launch_gemm_streamk(const GemmArgsT<R>& a, hipStream_t s) {
  const int first = streamk_first_tile(a);
  if (first < 0) return 1;
  if (a.epi && !gemm_streamk_carries(a)) return 1;   // the other fused epilogues live in the data-parallel kernels only
  int device = 0;
  GP_HIP(hipGetDevice(&device));
  SkPlan plan;
  void* fallback_ws = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    const SkKey key{device, a.M, a.N, a.K, a.out_lower, a.a_tri, a.b_tri, (a.reverse & 3) | (first ? 4 : 0), (int)sizeof(R)};
    auto itp = g_plans.find(key);
    if (itp == g_plans.end()) {
      SkPlan np;
      const int rc = build_plan(a, first, np);
      if (rc != 0) return rc;
      itp = g_plans.emplace(key, np).first;
    }
    plan = itp->second;
    if (!a.sk_ws) {
      void*& w = g_workspace[{device, s}];
      if (!w) GP_HIP(hipMalloc(&w, SK_WS_BYTES));
      fallback_ws = w;
    }
  }
  if (first > 0) {  // the full rounds, data-parallel
    GemmArgsT<R> head = a;
    head.tile = TILE;
    head.tile_limit = first;
    const int rc = launch_gemm_plain(head, s);
    if (rc != 0) return rc;
  }
  SkParams<R> p{};
  p.A = a.A; p.B = a.B; p.C = a.C; p.lda = a.lda; p.ldb = a.ldb; p.ldc = a.ldc; p.M = a.M; p.N = a.N;
  p.alpha = a.alpha; p.beta = a.beta; p.tiles = plan.tiles; p.ntiles = plan.ntiles; p.total = plan.total;
  p.per_block = plan.per_block; p.partial = (R*)(a.sk_ws ? a.sk_ws : fallback_ws);
  p.sumsq = a.sumsq; p.nt_all = (a.M / TILE) * (a.M / TILE + 1) / 2;
  dim3 grid(plan.blocks), block(GEMM_THREADS);
  const int sel = (a.a_kmajor ? 2 : 0) | (a.b_kmajor ? 1 : 0);
  if (a.epi == 0) switch (sel) {
    case 0: hipLaunchKernelGGL((gemm_streamk_kernel<R, false, false>), grid, block, 0, s, p); break;
    case 1: hipLaunchKernelGGL((gemm_streamk_kernel<R, false, true>), grid, block, 0, s, p); break;
    case 2: hipLaunchKernelGGL((gemm_streamk_kernel<R, true, false>), grid, block, 0, s, p); break;
    case 3: hipLaunchKernelGGL((gemm_streamk_kernel<R, true, true>), grid, block, 0, s, p); break;
  }
  if (a.epi == 0) {
    if (plan.nfix)
      hipLaunchKernelGGL(streamk_fixup_kernel<R>, dim3(plan.nfix, 32), dim3(256), 0, s, p, plan.fix_tile, plan.fix_ptr,
                         plan.fix_slot);
    GP_HIP(hipGetLastError());
    return 0;
  }
  if (a.epi == 2) {
    hipLaunchKernelGGL((gemm_streamk_kernel<R, false, true, 2>), grid, block, 0, s, p);
    if (plan.nfix)
      hipLaunchKernelGGL((streamk_fixup_kernel<R, 2>), dim3(plan.nfix, 32), dim3(256), 0, s, p, plan.fix_tile, plan.fix_ptr,
                         plan.fix_slot);
    GP_HIP(hipGetLastError());
    return 0;
  }
  return 1;
}

template int launch_gemm_streamk<double>(const GemmArgsT<double>&, hipStream_t);
template int launch_gemm_streamk<float>(const GemmArgsT<float>&, hipStream_t);

}  // namespace gpfit
