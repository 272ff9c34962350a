// Dev harness: ablations of the fp64 GEMM main loop on a plain 8192^3 problem (not shipped).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igaussian_processes_amd/csrc scripts/scratch/dev_gemm_abl.hip -o gpurun_tmp/abl
#include "gemm_core.h"
#include <cmath>
#include <cstdio>
#include <vector>
using namespace gpfit;

template <int ABL, bool AK, bool BK>
__global__ __launch_bounds__(256, 2) void k(const double* A, const double* B, double* C, int n, int K, int ld = 0, int mode = 0) {
  if (ld == 0) ld = n;
  __shared__ __attribute__((aligned(16))) double smem[4 * 16 * 128];
  const int tiles = n / 128;
  int ti = blockIdx.x / tiles, tj = blockIdx.x % tiles, kbeg = 0;
  if (mode == 1) { tj = blockIdx.x / tiles; ti = blockIdx.x % tiles; kbeg = tj * 128; }
  if (mode == 2) kbeg = ti * 128;
  v4d acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = acc_zero<double>();
  gemm_mainloop<double, AK, BK, false, 128, ABL>(A, ld, B, ld, n, n, ti * 128, tj * 128, kbeg, K, smem, acc);
  for_each_acc<double, 128>(acc, ti * 128, tj * 128, [&](int r, int c, double v) { C[(int64_t)r * ld + c] = v; });
}

template <bool AK, bool BK> void run_tri(const char* name, double* A, double* B, double* C, int n, int ld, int mode) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int nb = (n / 128) * (n / 128);
  k<0, AK, BK><<<nb, 256>>>(A, B, C, n, n, ld, mode);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0); k<0, AK, BK><<<nb, 256>>>(A, B, C, n, n, ld, mode); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
  }
  const double t = n / 128;
  const double flops = (mode == 0) ? 2.0 * n * n * n : 2.0 * 128 * 128 * 128 * t * (t * (t + 1) / 2);
  printf("%-44s %8.3f ms  %6.1f TF/s\n", name, best, flops / best / 1e9);
  fflush(stdout);
}

template <int ABL, bool AK, bool BK> void run(const char* name, double* A, double* B, double* C, int n) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int nb = (n / 128) * (n / 128);
  k<ABL, AK, BK><<<nb, 256>>>(A, B, C, n, n);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0); k<ABL, AK, BK><<<nb, 256>>>(A, B, C, n, n); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
  }
  double maxerr = 0;
  if (ABL == 0) {
    std::vector<double> hA((size_t)n * n), hB((size_t)n * n);
    hipMemcpy(hA.data(), A, sizeof(double) * n * n, hipMemcpyDeviceToHost);
    hipMemcpy(hB.data(), B, sizeof(double) * n * n, hipMemcpyDeviceToHost);
    const int pts[6][2] = {{0, 0}, {1, 130}, {127, 8191}, {4097, 77}, {8191, 8191}, {3000, 5001}};
    for (auto& pt : pts) {
      double ref = 0;
      for (int kq = 0; kq < n; ++kq) {
        const double a = AK ? hA[(size_t)kq * n + pt[0]] : hA[(size_t)pt[0] * n + kq];
        const double b = BK ? hB[(size_t)kq * n + pt[1]] : hB[(size_t)pt[1] * n + kq];
        ref += a * b;
      }
      double got;
      hipMemcpy(&got, C + (size_t)pt[0] * n + pt[1], sizeof(double), hipMemcpyDeviceToHost);
      maxerr = fmax(maxerr, fabs(got - ref));
    }
  }
  printf("%-44s %8.3f ms  %6.1f TF/s   maxerr %.2e\n", name, best, 2.0 * n * n * n / best / 1e9, maxerr);
  fflush(stdout);
}

int main() {
  const int n = 8192;
  double *A, *B, *C;
  hipMalloc(&A, sizeof(double) * n * n); hipMalloc(&B, sizeof(double) * n * n); hipMalloc(&C, sizeof(double) * n * n);
  std::vector<double> h((size_t)n * n);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  hipMemcpy(A, h.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 40503u + 17) % 977) / 977.0 - 0.5;
  hipMemcpy(B, h.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
  run<0, false, true>("full (A row-major, B k-major)", A, B, C, n);
  run<0, true, true>("full (both k-major)", A, B, C, n);
  run<0, false, false>("full (both k-contiguous)", A, B, C, n);
  run<1, false, true>("no barrier", A, B, C, n);
  run<2, false, true>("no DMA in loop", A, B, C, n);
  run<3, false, true>("no DMA, no barrier", A, B, C, n);
  run_tri<false, true>("4096 plain (A row, B kmajor), ld 8192", A, B, C, 4096, 8192, 0);
  run_tri<true, true>("4096 plain (both kmajor), ld 8192", A, B, C, 4096, 8192, 0);
  run_tri<false, true>("4096 b_tri col-major (A row, B kmajor)", A, B, C, 4096, 8192, 1);
  run_tri<true, true>("4096 b_tri col-major (both kmajor)", A, B, C, 4096, 8192, 1);
  run_tri<true, true>("4096 a_tri row-major (both kmajor)", A, B, C, 4096, 8192, 2);
  run_tri<true, true>("8192 b_tri col-major (both kmajor)", A, B, C, 8192, 8192, 1);
  run_tri<false, true>("2048 plain (A row, B kmajor), ld 8192", A, B, C, 2048, 8192, 0);
  return 0;
}
