// Phase timing inside the register-resident leaf (dev tool):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igaussian_processes_amd/csrc scripts/scratch/dev_leaf_time.hip -o /tmp/leaf_time && /tmp/leaf_time
// -DNO_STAMPS: the production kernel (no in-kernel stamps: each s_memtime sits on the critical path), timing only
#ifndef NO_STAMPS
#define GPFIT_LEAF_STAMPS 1
#endif
#ifdef MIN_STAMPS   // arrivals at B1 only: which side (pivot wave / tile waves) the other waits for
#define GPFIT_LEAF_STAMPS_MIN 1
#endif
#include "../../gaussian_processes_amd/csrc/chol_leaf_reg.hip"
#include <vector>
#include <cstdio>
#include <cmath>
namespace gpfit { void set_error(const std::string&) {} }
int main() {
  const int n = 128;
  std::vector<double> A(n * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i * n + j] = (i == j ? n : 0.0) + std::cos(0.37 * (i + 1) * (j + 1)) * 0.5 + std::cos(0.37 * (j + 1) * (i + 1)) * 0.5;
  double *dA, *dL, *dI; int* info;
  hipMalloc(&dA, n * n * 8); hipMalloc(&dL, n * n * 8); hipMalloc(&dI, n * n * 8); hipMalloc(&info, 16);
  hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice); hipMemset(info, 0, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) gpfit::launch_chol_leaf_reg<double>(dA, n, dL, n, dI, n, info, 0, 0);
  hipEventRecord(e0, 0);
  for (int it = 0; it < 100; ++it) gpfit::launch_chol_leaf_reg<double>(dA, n, dL, n, dI, n, info, 0, 0);
  hipEventRecord(e1, 0); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("back-to-back leaf launches: %.2f us each\n", ms * 10.0);
  {
    // correctness of the last launch: L L^T = A, L Li = I (host check, n = 128)
    std::vector<double> L(n * n), Li(n * n);
    hipMemcpy(L.data(), dL, n * n * 8, hipMemcpyDeviceToHost); hipMemcpy(Li.data(), dI, n * n * 8, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0, up = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
      double s1 = 0, s2 = 0;
      for (int k = 0; k < n; ++k) { s1 += L[i * n + k] * L[j * n + k]; s2 += L[i * n + k] * Li[k * n + j]; }
      e1 = std::fmax(e1, std::fabs(s1 - A[i * n + j])); e2 = std::fmax(e2, std::fabs(s2 - (i == j ? 1.0 : 0.0)));
      if (j > i) up = std::fmax(up, std::fmax(std::fabs(L[i * n + j]), std::fabs(Li[i * n + j])));
    }
    int h_info = -1; hipMemcpy(&h_info, info, 4, hipMemcpyDeviceToHost);
    printf("max |L L^T - A| %.3e  max |L Li - I| %.3e  max strict-upper entry %.3e  info %d\n", e1, e2, up, h_info);
  }
#ifdef NO_STAMPS
  return 0;
#else
  long long st[72];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(gpfit::g_leaf_stamps), sizeof(st));
  const double t0 = (double)st[64];
  printf("panels start at 0, kernel end %.0f cycles\n", (double)st[65] - t0);
  const char* names[7] = {"start", "rows-read", "pivots+inv", "written", "barrier1", "solve+b2", "updates+b3"};
  for (int kb = 0; kb < 8; ++kb) {
    printf("panel %d:", kb);
    for (int ph = 0; ph < 7; ++ph) printf(" %s %.0f", names[ph], (double)st[kb * 8 + ph] - t0);
    printf("\n");
  }
  long long we[32];
  hipMemcpyFromSymbol(we, HIP_SYMBOL(gpfit::g_leaf_wave_end), sizeof(we));
#ifdef MIN_STAMPS
  for (int kb = 0; kb < 8; ++kb)
    printf("panel %d: arrivals at B1(%d): pivot wave %.0f (its chain of panel %d written), tile waves %.0f %.0f %.0f %.0f (update pass of panel %d done)\n", kb, kb + 1,
           kb + 1 < 8 ? (double)st[(kb + 1) * 8 + 3] - t0 : 0.0, kb + 1, (double)we[kb * 4] - t0, (double)we[kb * 4 + 1] - t0, (double)we[kb * 4 + 2] - t0,
           (double)we[kb * 4 + 3] - t0, kb);
  return 0;
#endif
  for (int kb = 0; kb < 8; ++kb)
    printf("panel %d: update pass of tile waves 0..3 ends %.0f %.0f %.0f %.0f after B2 (%.0f)\n", kb, (double)we[kb * 4] - (double)st[kb * 8 + 5],
           (double)we[kb * 4 + 1] - (double)st[kb * 8 + 5], (double)we[kb * 4 + 2] - (double)st[kb * 8 + 5], (double)we[kb * 4 + 3] - (double)st[kb * 8 + 5],
           (double)st[kb * 8 + 5] - t0);
  return 0;
#endif
}
