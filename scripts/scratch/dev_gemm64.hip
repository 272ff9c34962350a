// Dev harness (not shipped): where does the 64-tile fp64 GEMM lose its time on 2048-sized problems?
// Dense C_b = A_b B_b^T (A row-major, B k-major or k-contiguous) for nb problems of size n, tile T, NS stages,
// ablations of the main loop (1 no barrier, 2 no DMA in the loop), occupancy 1..4 workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igaussian_processes_amd/csrc scripts/scratch/dev_gemm64.hip -o gpurun_tmp/gemm64
#include "gemm_core.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace gpfit;

template <int T, int NS, int ABL, bool BK, int OCC>
__global__ __launch_bounds__(256, OCC) void k(const double* A, const double* B, double* C, int n, int64_t stride, int walk) {
  __shared__ __attribute__((aligned(16))) double smem[2 * NS * 16 * T];
  const int tiles = n / T;
  int ti, tj;
  if (walk == 0) { ti = blockIdx.x / tiles; tj = blockIdx.x % tiles; }
  else {
    // XCD-aware: ids = x (mod 8) go to XCD x; give XCD x an (tiles/2) x (tiles/4) patch of the output, walked row-major
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int pr = tiles / 2, pc = tiles / 4;   // patch rows / cols
    ti = (x >> 2) * pr + q / pc;
    tj = (x & 3) * pc + q % pc;
  }
  A += stride * blockIdx.y; B += stride * blockIdx.y; C += stride * blockIdx.y;
  v4d acc[T / 32][T / 32];
  for (int i = 0; i < T / 32; ++i) for (int j = 0; j < T / 32; ++j) acc[i][j] = acc_zero<double>();
  gemm_mainloop<double, false, BK, false, T, ABL, NS>(A, n, B, n, n, n, ti * T, tj * T, 0, n, smem, acc);
  for_each_acc<double, T>(acc, ti * T, tj * T, [&](int r, int c, double v) { C[(int64_t)r * n + c] = v; });
}


// ---- harness-local copy of gemm_core.h's DMA main loop with variation points (VAR bits):
//   1  the DMA pieces of tile s+D are issued one per sub-step instead of all at the top of the K step
//   2  (with 1) staggered by wave: wave w issues piece (kk + w) % OPS at sub-step kk
template <typename R, bool A_KMAJOR, bool B_KMAJOR, int T, int NS, int VAR>
__device__ __forceinline__ void mainloop_var(const R* __restrict__ A, int64_t lda, const R* __restrict__ B, int64_t ldb,
                                             int row0, int col0, int kbeg, int kend, R* smem,
                                             typename Real<R>::acc_t (&acc)[T / 32][T / 32]) {
  constexpr int MI = T / 32, WT = T / 2, KT = Real<R>::KT, LT = KT * T, NKK = KT / 4;
  constexpr int D = NS - 1;
  constexpr int OPS = 2 * MI;
  static_assert(OPS % NKK == 0 || NKK % OPS == 0, "pieces per sub-step");
  constexpr int PPS = OPS >= NKK ? OPS / NKK : 1;   // pieces per sub-step
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
  da.init(A, lda, row0, kbeg, wave, lane);
  db.init(B, ldb, col0, kbeg, wave, lane);
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  constexpr uint32_t LTB = LT * sizeof(R);
  auto issue = [&](int stage) {
    const uint32_t img = lds0 + (uint32_t)stage * 2u * LTB;
    da.issue(img);
    db.issue(img + LTB);
    da.advance();
    db.advance();
  };
  auto piece = [&](int stage, int i) {   // i in [0, OPS): A pieces first
    const uint32_t img = lds0 + (uint32_t)stage * 2u * LTB;
    if (i < MI) da.issue_one(img, i);
    else db.issue_one(img + LTB, i - MI);
  };
  __syncthreads();
#pragma unroll
  for (int t = 0; t < D; ++t)
    if (t < ntile) issue(t);
  if (ntile >= D) dma_wait_barrier<(D - 1) * OPS>();
  else dma_wait_barrier<0>();
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
  int cs = 0;
  for (int s = 0; s < ntile; ++s) {
    const bool more = (s + 1) < ntile;
    const int nx = (cs + 1 == NS) ? 0 : cs + 1;
    const R* Sc = smem + cs * 2 * LT;
    const R* Sn = smem + nx * 2 * LT;
    const bool fetch = s + D < ntile;
    const int fstage = (cs + D) % NS;
    if (!(VAR & 1) && fetch) issue(fstage);
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if ((VAR & 1) && fetch) {
        if (OPS >= NKK) {
#pragma unroll
          for (int q = 0; q < PPS; ++q) piece(fstage, kk * PPS + q);
        } else if (kk % (NKK / OPS) == 0) piece(fstage, kk / (NKK / OPS));
        if (kk == NKK - 1) { da.advance(); db.advance(); }
      }
      if (kk + 1 < NKK) {
        frag(Sc, kk + 1, fa[nxt], fb[nxt]);
      } else if (more) {
        if (s + D < ntile) dma_wait_barrier<(D - 1) * OPS>();
        else dma_wait_barrier<0>();
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

template <int T, int NS, int VAR, bool BK, int OCC>
__global__ __launch_bounds__(256, OCC) void kv(const double* A, const double* B, double* C, int n, int64_t stride, int walk) {
  __shared__ __attribute__((aligned(16))) double smem[2 * NS * 16 * T];
  const int tiles = n / T;
  const int ti = blockIdx.x / tiles, tj = blockIdx.x % tiles;
  A += stride * blockIdx.y; B += stride * blockIdx.y; C += stride * blockIdx.y;
  v4d acc[T / 32][T / 32];
  for (int i = 0; i < T / 32; ++i) for (int j = 0; j < T / 32; ++j) acc[i][j] = acc_zero<double>();
  mainloop_var<double, false, BK, T, NS, VAR>(A, n, B, n, ti * T, tj * T, 0, n, smem, acc);
  for_each_acc<double, T>(acc, ti * T, tj * T, [&](int r, int c, double v) { C[(int64_t)r * n + c] = v; });
}

template <int T, int NS, int ABL, bool BK, int OCC>
void run(const char* name, double* A, double* B, double* C, int n, int nb, int walk = 0) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid((n / T) * (n / T), nb);
  const int64_t stride = (int64_t)n * n;
  k<T, NS, ABL, BK, OCC><<<grid, 256>>>(A, B, C, n, stride, walk);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int it = 0; it < 5; ++it) {
    hipEventRecord(e0); k<T, NS, ABL, BK, OCC><<<grid, 256>>>(A, B, C, n, stride, walk); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
  }
  double maxerr = -1;
  if (ABL == 0) {
    maxerr = 0;
    std::vector<double> hA((size_t)n * n), hB((size_t)n * n);
    const int b = nb - 1;
    hipMemcpy(hA.data(), A + stride * b, sizeof(double) * n * n, hipMemcpyDeviceToHost);
    hipMemcpy(hB.data(), B + stride * b, sizeof(double) * n * n, hipMemcpyDeviceToHost);
    const int pts[5][2] = {{0, 0}, {1, 130}, {127, n - 1}, {n - 1, n - 1}, {n / 2 + 3, 77}};
    for (auto& pt : pts) {
      double ref = 0;
      for (int kq = 0; kq < n; ++kq) ref += hA[(size_t)pt[0] * n + kq] * (BK ? hB[(size_t)kq * n + pt[1]] : hB[(size_t)pt[1] * n + kq]);
      double got;
      hipMemcpy(&got, C + stride * b + (size_t)pt[0] * n + pt[1], sizeof(double), hipMemcpyDeviceToHost);
      maxerr = fmax(maxerr, fabs(got - ref));
    }
  }
  printf("%-52s n %5d nb %d: %8.1f us  %6.1f TF/s   maxerr %.1e\n", name, n, nb, best * 1e3, 2.0 * n * n * n * nb / best / 1e9, maxerr);
  fflush(stdout);
}

template <int T, int NS, int VAR, bool BK, int OCC>
void runv(const char* name, double* A, double* B, double* C, int n, int nb) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid((n / T) * (n / T), nb);
  const int64_t stride = (int64_t)n * n;
  kv<T, NS, VAR, BK, OCC><<<grid, 256>>>(A, B, C, n, stride, 0);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int it = 0; it < 5; ++it) {
    hipEventRecord(e0); kv<T, NS, VAR, BK, OCC><<<grid, 256>>>(A, B, C, n, stride, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
  }
  double maxerr = 0;
  std::vector<double> hA((size_t)n * n), hB((size_t)n * n);
  const int b = nb - 1;
  hipMemcpy(hA.data(), A + stride * b, sizeof(double) * n * n, hipMemcpyDeviceToHost);
  hipMemcpy(hB.data(), B + stride * b, sizeof(double) * n * n, hipMemcpyDeviceToHost);
  const int pts[5][2] = {{0, 0}, {1, 130}, {127, n - 1}, {n - 1, n - 1}, {n / 2 + 3, 77}};
  for (auto& pt : pts) {
    double ref = 0;
    for (int kq = 0; kq < n; ++kq) ref += hA[(size_t)pt[0] * n + kq] * (BK ? hB[(size_t)kq * n + pt[1]] : hB[(size_t)pt[1] * n + kq]);
    double got;
    hipMemcpy(&got, C + stride * b + (size_t)pt[0] * n + pt[1], sizeof(double), hipMemcpyDeviceToHost);
    maxerr = fmax(maxerr, fabs(got - ref));
  }
  printf("%-52s n %5d nb %d: %8.1f us  %6.1f TF/s   maxerr %.1e\n", name, n, nb, best * 1e3, 2.0 * n * n * n * nb / best / 1e9, maxerr);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 2048;
  const int NB = 4;
  double *A, *B, *C;
  const size_t tot = (size_t)n * n * NB;
  hipMalloc(&A, sizeof(double) * tot); hipMalloc(&B, sizeof(double) * tot); hipMalloc(&C, sizeof(double) * tot);
  std::vector<double> h(tot);
  for (size_t i = 0; i < tot; ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  hipMemcpy(A, h.data(), sizeof(double) * tot, hipMemcpyHostToDevice);
  for (size_t i = 0; i < tot; ++i) h[i] = (double)((i * 40503u + 17) % 977) / 977.0 - 0.5;
  hipMemcpy(B, h.data(), sizeof(double) * tot, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep)
  for (int nb : {1, 4}) {
    runv<64, 2, 0, true, 4>("var0 T64 NS2 occ4 (copy of the product loop)", A, B, C, n, nb);
    run<64, 2, 0, true, 4>("T64 NS2 occ4 (B k-major)", A, B, C, n, nb);
    runv<64, 2, 0, true, 4>("var0 T64 NS2 occ4 (copy of the product loop)", A, B, C, n, nb);
    run<64, 2, 0, true, 4>("T64 NS2 occ4 (B k-major)", A, B, C, n, nb);
    run<128, 2, 0, true, 2>("T128 NS2 occ2 (B k-major)", A, B, C, n, nb);
    runv<128, 2, 0, true, 2>("var0 T128 NS2 occ2 (copy)", A, B, C, n, nb);
    runv<64, 2, 1, true, 4>("var1 T64 NS2 occ4 pieces spread", A, B, C, n, nb);
    runv<64, 2, 0, true, 4>("var0 T64 NS2 occ4 (copy of the product loop)", A, B, C, n, nb);
    run<64, 4, 0, true, 2>("T64 NS4 occ2 (B k-major)", A, B, C, n, nb);
    run<64, 2, 3, true, 4>("T64 NS2 occ4 no DMA, no barrier", A, B, C, n, nb);
  }
  return 0;
}
