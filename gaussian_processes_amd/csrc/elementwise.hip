// Element-wise / reduction kernels of the GP fit path (everything that is not an MFMA GEMM
// or the Cholesky leaf).  All of them are HBM- or latency-bound and tiny next to the N^3
// work; they are written for coalesced access and deterministic (atomic-free) reductions.
#include "kernels.h"

namespace gpfit {

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  return v;
}

// Sum over the whole block; result valid in every thread.  sh: >= 17 doubles.
__device__ __forceinline__ double block_sum(double v, double* sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  if (wave == 0) {
    double t = (lane < nw) ? sh[lane] : 0.0;
    t = wave_sum(t);
    if (lane == 0) sh[16] = t;
  }
  __syncthreads();
  return sh[16];
}

// torch.linspace(-1, 1, n)[i] in fp64: two-sided fused form (matches ATen bit for bit;
// the pixel mesh of reference utils.py:876).
__device__ __forceinline__ double lin_pm1(int i, int n) {
  if (n <= 1) return -1.0;
  const double step = 2.0 / (double)(n - 1);
  return (i < n / 2) ? fma(step, (double)i, -1.0) : fma(-step, (double)(n - 1 - i), 1.0);
}

// ------------------------------------------------------------------ localker
template <typename R>
__global__ void localker_kernel(Theta th, const int* __restrict__ pix, int d, int dp, int n_rows, int n_cols,
                                R* __restrict__ C, int64_t ldc, R* __restrict__ dC) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= dp || j >= dp) return;
  if (i >= d || j >= d) {
    C[(int64_t)i * ldc + j] = (R)0;
    return;
  }
  const int pi = pix[i], pj = pix[j];
  const double xi = lin_pm1(pi % n_cols, n_cols), yi = lin_pm1(pi / n_cols, n_rows);
  const double xj = lin_pm1(pj % n_cols, n_cols), yj = lin_pm1(pj / n_cols, n_rows);
  const double dxi = xi - th.eps0x, dyi = yi - th.eps0y, dxj = xj - th.eps0x, dyj = yj - th.eps0y;
  const double lai = -th.eb * (dxi * dxi + dyi * dyi);  // utils.py:880
  const double laj = -th.eb * (dxj * dxj + dyj * dyj);
  const double ai = exp(lai), aj = exp(laj);            // :881
  const double ex = xj - xi, ey = yj - yi;
  const double ls = -th.er * (ex * ex + ey * ey);       // :890
  const double es = exp(ls);                            // :892
  const double cij = th.amp * ai * es * aj;             // :895
  const double cji = th.amp * aj * es * ai;
  const double c = (cij + cji) / 2.0;                   // :898
  C[(int64_t)i * ldc + j] = (R)c;
  if (dC) {
    const int64_t dd = (int64_t)d * d, o = (int64_t)i * d + j;
    dC[o] = (R)(c / th.amp);                                           // Amp        :902
    dC[dd + o] = (R)(c * (lai + laj));                                 // -2log2beta :907
    dC[2 * dd + o] = (R)(c * ls);                                      // -log2rho2  :909
    dC[3 * dd + o] = (R)(2.0 * th.eb * c * (xi + xj - 2.0 * th.eps0x));  // eps_0x   :904
    dC[4 * dd + o] = (R)(2.0 * th.eb * c * (yi + yj - 2.0 * th.eps0y));  // eps_0y   :905
  }
}

template <typename R>
int launch_localker(const Theta& th, const int* pix, int d, int dp, int n_rows, int n_cols, R* C, int64_t ldc,
                    R* dC, hipStream_t s) {
  dim3 block(32, 8), grid((dp + 31) / 32, (dp + 7) / 8);
  hipLaunchKernelGGL(localker_kernel<R>, grid, block, 0, s, th, pix, d, dp, n_rows, n_cols, C, ldc, dC);
  GP_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ gather / transpose
template <typename R>
__global__ void gather_kernel(const R* __restrict__ X, int64_t ldx, int n, const int* __restrict__ pix, int d,
                              R* __restrict__ Xt, int64_t ldt, R* __restrict__ Xm, int64_t ldm) {
  __shared__ R tile[32][33];
  const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y;  // 32 x 8
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int nn = n0 + ty + 8 * r, k = k0 + tx;
    R v = 0;
    if (nn < n && k < d) v = X[(int64_t)nn * ldx + (pix ? pix[k] : k)];
    tile[ty + 8 * r][tx] = v;
    if (Xm) Xm[(int64_t)nn * ldm + k] = v;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = k0 + ty + 8 * r, nn = n0 + tx;
    Xt[(int64_t)k * ldt + nn] = tile[tx][ty + 8 * r];
  }
}

template <typename R>
int launch_gather(const R* X, int64_t ldx, int n, const int* pix, int d, int dp, int np, R* Xt, int64_t ldt, R* Xm,
                  int64_t ldm, hipStream_t s) {
  if (dp % 32 || np % 32) {
    set_error("launch_gather: padded extents must be multiples of 32");
    return -3;
  }
  hipLaunchKernelGGL(gather_kernel<R>, dim3(dp / 32, np / 32), dim3(32, 8), 0, s, X, ldx, n, pix, d, Xt, ldt, Xm,
                     ldm);
  GP_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ q / Kvec
// One block = 32 samples x 8 slices of the pixel range: a thread sums every eighth pixel of its sample (coalesced over
// the samples), the eight partial sums meet in LDS in a fixed order.  (One thread per sample walked all dp pixels
// with one load in flight: 37 us at N = 4096, 67 us at N = 8192 -- latency, not bandwidth.)
template <typename R>
__global__ __launch_bounds__(256) void qvec_kernel(const R* __restrict__ Xt, const R* __restrict__ XCt, int64_t ld, int dp,
                                                   int n, int np, double s0sq, R* __restrict__ Kvec, R* __restrict__ q) {
  __shared__ double part[8][33];
  const int li = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + li;
  double h = 0.0;
  if (i < n) {
#pragma unroll 4
    for (int k = sl; k < dp; k += 8) h += (double)Xt[(int64_t)k * ld + i] * (double)XCt[(int64_t)k * ld + i];
  }
  part[sl][li] = h;
  __syncthreads();
  if (sl != 0 || i >= np) return;
  if (i >= n) {
    Kvec[i] = (R)1;
    q[i] = (R)1;
    return;
  }
  double t = part[0][li];
#pragma unroll
  for (int z = 1; z < 8; ++z) t += part[z][li];
  const double kv = t + s0sq;  // utils.py:1029
  Kvec[i] = (R)kv;
  q[i] = (R)sqrt(kv);          // utils.py:978
}

template <typename R>
int launch_qvec(const R* Xt, const R* XCt, int64_t ld, int dp, int n, int np, double s0sq, R* Kvec, R* q,
                hipStream_t s) {
  hipLaunchKernelGGL(qvec_kernel<R>, dim3((np + 31) / 32), dim3(256), 0, s, Xt, XCt, ld, dp, n, np, s0sq, Kvec, q);
  GP_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ group housekeeping (kernels.h)
template <typename R>
__global__ __launch_bounds__(256) void group_prepare_kernel(GroupPrepT<R> g) {
  const int u = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < g.np) g.mpad[u][i] = (i < g.n) ? g.m[u][i] : (R)0;
  if (blockIdx.x == 0) {
    for (int k = threadIdx.x; k < g.d[u]; k += 256) g.pix[u][k] = g.pix_host[u][k];
    if (threadIdx.x < 4) g.info[u][threadIdx.x] = 0;
  }
}

template <typename R>
int launch_group_prepare(const GroupPrepT<R>& g, hipStream_t s) {
  hipLaunchKernelGGL(group_prepare_kernel<R>, dim3((g.np + 255) / 256, g.n_units), dim3(256), 0, s, g);
  GP_HIP(hipGetLastError());
  return 0;
}

__global__ void group_collect_kernel(GroupCollectT g) {
  const int u = blockIdx.x, t = threadIdx.x;
  if (t < 64) g.scal_host[u][t] = g.scal[u][t];
  else if (t < 68) g.info_host[u][t - 64] = g.info[u][t - 64];
  __threadfence_system();
}

int launch_group_collect(const GroupCollectT& g, hipStream_t s) {
  hipLaunchKernelGGL(group_collect_kernel, dim3(g.n_units), dim3(128), 0, s, g);
  GP_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ pack / symmetrize
// (a tile is copied by eight blocks, sixteen rows each: with one block per tile a 640 x 640 matrix had 15 blocks on
// the chip and the copy took 21-33 us; 64 us at N = 4096)
template <typename R>
__global__ __launch_bounds__(256) void pack_lower_kernel(const R* __restrict__ src, int64_t lds, int n, R* __restrict__ dst,
                                                         int64_t ldd) {
  const int tj = blockIdx.x, ti = blockIdx.y;
  if (tj > ti) return;
  const int r0 = ti * TILE + 16 * blockIdx.z, c0 = tj * TILE;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int e = threadIdx.x + 256 * it;
    const int i = r0 + (e >> 7), j = c0 + (e & 127);
    R v;
    if (i < n && j < n) v = src[(int64_t)i * lds + j];
    else v = (i == j) ? (R)1 : (R)0;
    dst[(int64_t)i * ldd + j] = v;
  }
}

template <typename R>
int launch_pack_lower(const R* src, int64_t lds, int n, R* dst, int64_t ldd, int np, hipStream_t s) {
  hipLaunchKernelGGL(pack_lower_kernel<R>, dim3(np / TILE, np / TILE, TILE / 16), dim3(256), 0, s, src, lds, n, dst, ldd);
  GP_HIP(hipGetLastError());
  return 0;
}

template <typename R>
__global__ void symmetrize_kernel(R* __restrict__ A, int64_t lda, int n) {
  __shared__ R tile[32][33];
  const int tj = blockIdx.x, ti = blockIdx.y;
  if (tj > ti) return;
  const int tx = threadIdx.x, ty = threadIdx.y;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = ti * 32 + ty + 8 * r, j = tj * 32 + tx;
    tile[ty + 8 * r][tx] = (i < n && j < n) ? A[(int64_t)i * lda + j] : (R)0;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    // element (jj, ii) of the upper triangle <- (ii, jj) of the lower one
    const int jj = tj * 32 + ty + 8 * r, ii = ti * 32 + tx;
    if (ii < n && jj < n && ii > jj) A[(int64_t)jj * lda + ii] = tile[tx][ty + 8 * r];
  }
}

template <typename R>
int launch_symmetrize(R* A, int64_t lda, int n, hipStream_t s) {
  const int t = (n + 31) / 32;
  hipLaunchKernelGGL(symmetrize_kernel<R>, dim3(t, t), dim3(32, 8), 0, s, A, lda, n);
  GP_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ small reductions
template <typename R>
__global__ void logdet_kernel(const R* __restrict__ L0, double* __restrict__ out0, const R* __restrict__ L1,
                              double* __restrict__ out1, int64_t ldl, int n) {
  __shared__ double sh[17];
  const R* __restrict__ L = blockIdx.x == 0 ? L0 : L1;
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) v += log((double)L[(int64_t)i * ldl + i]);
  v = block_sum(v, sh);
  if (threadIdx.x == 0) (blockIdx.x == 0 ? out0 : out1)[0] = 2.0 * v;  // utils.py:1278
}

template <typename R>
int launch_logdet(const R* L, int64_t ldl, int n, double* out, hipStream_t s) {
  hipLaunchKernelGGL(logdet_kernel<R>, dim3(1), dim3(1024), 0, s, L, out, L, out, ldl, n);
  GP_HIP(hipGetLastError());
  return 0;
}

// two factors of the same size in one launch
template <typename R>
int launch_logdet_pair(const R* L0, double* out0, const R* L1, double* out1, int64_t ldl, int n, hipStream_t s) {
  hipLaunchKernelGGL(logdet_kernel<R>, dim3(2), dim3(1024), 0, s, L0, out0, L1, out1, ldl, n);
  GP_HIP(hipGetLastError());
  return 0;
}

template <typename R>
__global__ void sum_kernel(const R* __restrict__ x, int n, double scale, double* __restrict__ out) {
  __shared__ double sh[17];
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) v += (double)x[i];
  v = block_sum(v, sh);
  if (threadIdx.x == 0) out[0] = scale * v;
}

template <typename R>
__global__ void frob_tile_kernel(const R* __restrict__ T, int64_t ldt, double* __restrict__ partial) {
  __shared__ double sh[17];
  int ti, tj;
  {
    const int t = blockIdx.x;
    int i = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= t) ++i;
    while (i * (i + 1) / 2 > t) --i;
    ti = i;
    tj = t - i * (i + 1) / 2;
  }
  const R* base = T + (int64_t)ti * TILE * ldt + tj * TILE;
  double v = 0.0;
#pragma unroll 8
  for (int e = threadIdx.x; e < TILE * TILE; e += 256) {
    const double x = (double)base[(int64_t)(e >> 7) * ldt + (e & 127)];
    v += x * x;
  }
  v = block_sum(v, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = v;
}

template <typename R>
int launch_frob_lower(const R* T, int64_t ldt, int np, double* out, double* partial, hipStream_t s) {
  const int t = np / TILE, nt = t * (t + 1) / 2;
  hipLaunchKernelGGL(frob_tile_kernel<R>, dim3(nt), dim3(256), 0, s, T, ldt, partial);
  hipLaunchKernelGGL(sum_kernel<double>, dim3(1), dim3(1024), 0, s, partial, nt, 1.0, out);
  GP_HIP(hipGetLastError());
  return 0;
}

// out = sum of the nt per-tile sums of squares a GEMM with the tile-norm epilogue left in partial
int launch_frob_finish(const double* partial, int nt, double* out, hipStream_t s) {
  hipLaunchKernelGGL(sum_kernel<double>, dim3(1), dim3(1024), 0, s, partial, nt, 1.0, out);
  GP_HIP(hipGetLastError());
  return 0;
}

template <typename R>
__global__ void trmv_lower_kernel(const R* __restrict__ L, int64_t ldl, int np, const R* __restrict__ x,
                                  R* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= np) return;
  const R* row = L + (int64_t)i * ldl;
  double v = 0.0;
#pragma unroll 8
  for (int j = lane; j <= i; j += 64) v += (double)row[j] * (double)x[j];
  v = wave_sum(v);
  if (lane == 0) y[i] = (R)v;
}

template <typename R>
int launch_trmv_lower(const R* L, int64_t ldl, int np, const R* x, R* y, hipStream_t s) {
  hipLaunchKernelGGL(trmv_lower_kernel<R>, dim3((np + 3) / 4), dim3(256), 0, s, L, ldl, np, x, y);
  GP_HIP(hipGetLastError());
  return 0;
}

// z_j = sum_{i >= j} L[i][j] x_i : block = 64 columns x one chunk of TRMV_ROWS rows
template <typename R>
__global__ void trmv_lower_t_kernel(const R* __restrict__ L, int64_t ldl, int np, const R* __restrict__ x,
                                    double* __restrict__ partial) {
  __shared__ double sh[4][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + c;
  const int i0 = blockIdx.y * TRMV_ROWS, i1 = min(np, i0 + TRMV_ROWS);
  double v = 0.0;
  if (i1 > blockIdx.x * 64) {
#pragma unroll 8
    for (int i = i0 + rl; i < i1; i += 4)
      if (i >= j) v += (double)L[(int64_t)i * ldl + j] * (double)x[i];
  }
  sh[rl][c] = v;
  __syncthreads();
  if (rl == 0) partial[(int64_t)blockIdx.y * np + j] = sh[0][c] + sh[1][c] + sh[2][c] + sh[3][c];
}

template <typename RI, typename RO>
__global__ void reduce_slices_kernel(const RI* __restrict__ src, int64_t stride, int nslice, RO* __restrict__ dst,
                                     int64_t count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double v = 0.0;
  for (int z = 0; z < nslice; ++z) v += (double)src[(int64_t)z * stride + i];
  dst[i] = (RO)v;
}

template <typename RI, typename RO>
int launch_reduce_slices(const RI* src, int64_t slice_stride, int nslice, RO* dst, int64_t count, hipStream_t s) {
  hipLaunchKernelGGL((reduce_slices_kernel<RI, RO>), dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, src,
                     slice_stride, nslice, dst, count);
  GP_HIP(hipGetLastError());
  return 0;
}

template <typename R>
int launch_trmv_lower_t(const R* L, int64_t ldl, int np, const R* x, R* z, double* partial, hipStream_t s) {
  const int chunks = (np + TRMV_ROWS - 1) / TRMV_ROWS;
  hipLaunchKernelGGL(trmv_lower_t_kernel<R>, dim3(np / 64, chunks), dim3(256), 0, s, L, ldl, np, x, partial);
  GP_HIP(hipGetLastError());
  return launch_reduce_slices(partial, np, chunks, z, np, s);
}

template <typename R>
__global__ void dot_kernel(const R* __restrict__ x, const R* __restrict__ y, int n, double* __restrict__ out) {
  __shared__ double sh[17];
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) v += (double)x[i] * (double)y[i];
  v = block_sum(v, sh);
  if (threadIdx.x == 0) out[0] = v;
}

template <typename R>
int launch_dot(const R* x, const R* y, int n, double* out, hipStream_t s) {
  hipLaunchKernelGGL(dot_kernel<R>, dim3(1), dim3(1024), 0, s, x, y, n, out);
  GP_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ moments / rate / likelihood
// One thread per observation over n / 256 workgroups; the three sums go through per-workgroup partials that the last
// workgroup to finish (a ticket) adds up in workgroup order -- deterministic, and one launch.  (A single workgroup
// walking all n observations, two strided diagonal reads and an acos / exp each, took 25 us at N = 4096 and 51 us at
// N = 8192.)
template <typename R>
__global__ __launch_bounds__(256) void moments_kernel(const R* __restrict__ Kvec, const R* __restrict__ q,
                                                      const R* __restrict__ Cos, int64_t ldc, const R* __restrict__ V,
                                                      int64_t ldv, const R* __restrict__ m, const R* __restrict__ r, int n,
                                                      double A, double lambda0, R* __restrict__ lam_m,
                                                      R* __restrict__ lam_var, R* __restrict__ f, R* __restrict__ wl,
                                                      double* __restrict__ scal, double* __restrict__ part,
                                                      int* __restrict__ ticket) {
  __shared__ double sh[17];
  __shared__ bool last;
  double s_rm = 0.0, s_r = 0.0, s_f = 0.0;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const double c = (double)Cos[(int64_t)i * ldc + i];
    const double delta = acos(c);
    const double J = (sqrt(1.0 - c * c) + PI32 * c - delta * c) / PI32;
    const double qi = (double)q[i];
    const double kii = qi * qi * J;                         // K~_ii as the Gram kernel wrote it
    const double lv = (double)Kvec[i] - kii + (double)V[(int64_t)i * ldv + i];  // utils.py:1101, a = B, full rank
    const double lm = (double)m[i];                         // utils.py:1090
    const double fi = exp(A * lm + 0.5 * A * A * lv + lambda0);  // utils.py:1138
    const double g = 1.0 - J - (PI32 - delta) * (1.0 - c) / PI32;
    lam_m[i] = (R)lm;
    lam_var[i] = (R)lv;
    f[i] = (R)fi;
    wl[i] = (R)(-0.5 * A * A * fi * g);
    s_rm = (double)r[i] * lm;
    s_r = (double)r[i];
    s_f = fi;
  }
  s_rm = block_sum(s_rm, sh);
  s_r = block_sum(s_r, sh);
  s_f = block_sum(s_f, sh);
  if (threadIdx.x == 0) {
    part[blockIdx.x * 3 + 0] = s_rm;
    part[blockIdx.x * 3 + 1] = s_r;
    part[blockIdx.x * 3 + 2] = s_f;
    __threadfence();                                    // partials visible before the ticket
    last = (atomicAdd(ticket, 1) == (int)gridDim.x - 1);
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (threadIdx.x < 3) {
    double v = 0.0;
    for (int b = 0; b < (int)gridDim.x; ++b)
      v += __hip_atomic_load(&part[b * 3 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    scal[threadIdx.x] = v;
  }
  if (threadIdx.x == 0) *ticket = 0;
}

// part: >= 3 * ceil(n / 256) doubles of scratch; ticket: a device int that is 0 between calls
template <typename R>
int launch_moments(const R* Kvec, const R* q, const R* Cos, int64_t ldc, const R* V, int64_t ldv, const R* m,
                   const R* r, int n, double A, double lambda0, R* lam_m, R* lam_var, R* f, R* wl, double* scal,
                   double* part, int* ticket, hipStream_t s) {
  hipLaunchKernelGGL(moments_kernel<R>, dim3((n + 255) / 256), dim3(256), 0, s, Kvec, q, Cos, ldc, V, ldv, m, r, n, A,
                     lambda0, lam_m, lam_var, f, wl, scal, part, ticket);
  GP_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ adjoint pass (64 x 64 tiles)
template <typename R>
__global__ __launch_bounds__(256) void adjoint_kernel(const R* __restrict__ W, const R* __restrict__ Cos, int64_t ld,
                                                      const R* __restrict__ b, const R* __restrict__ q, int n,
                                                      int np, R* __restrict__ Aout, double* __restrict__ upart,
                                                      double* __restrict__ vpart, double* __restrict__ sumA_part) {
  __shared__ R tA[64][65];
  __shared__ double colsum[4][64];
  __shared__ double sh[17];
  int ti, tj;
  {
    const int t = blockIdx.x;
    int i = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= t) ++i;
    while (i * (i + 1) / 2 > t) --i;
    ti = i;
    tj = t - i * (i + 1) / 2;
  }
  const bool diag = (ti == tj);
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // ty = wave
  const int j = tj * 64 + tx;
  const double qj = (double)q[j], bj = (double)b[j];
  double csum = 0.0, asum = 0.0;
  for (int rr = ty; rr < 64; rr += 4) {
    const int i = ti * 64 + rr;
    double aw = 0.0, bm = 0.0;
    if (i < n && j < n) {
      // canonical storage is the lower triangle: in a diagonal tile read (j,i) when i < j
      const int64_t o = (diag && i < j) ? ((int64_t)j * ld + i) : ((int64_t)i * ld + j);
      const double w = (double)W[o] - 0.5 * (double)b[i] * bj;
      const double c = (double)Cos[o];
      aw = w * (PI32 - acos(c)) / PI32;
      bm = w * sqrt(1.0 - c * c) / PI32;
    }
    tA[rr][tx] = (R)aw;
    Aout[(int64_t)i * ld + j] = (R)aw;
    asum += aw;
    // row sum over the 64 columns of this tile (one wave holds a whole row)
    const double rs = wave_sum(bm * qj);
    if (tx == 0) upart[(int64_t)tj * np + i] = rs;
    csum += bm * (double)q[i];
  }
  colsum[ty][tx] = csum;
  __syncthreads();
  if (!diag) {
    if (ty == 0) vpart[(int64_t)ti * np + j] = colsum[0][tx] + colsum[1][tx] + colsum[2][tx] + colsum[3][tx];
    // mirrored tile: Aout[j][i] = Aw[i][j]
    for (int rr = ty; rr < 64; rr += 4) {
      const int jj = tj * 64 + rr, ii = ti * 64 + tx;
      Aout[(int64_t)jj * ld + ii] = tA[tx][rr];
    }
  }
  asum = block_sum(asum, sh);
  if (threadIdx.x == 0) sumA_part[blockIdx.x] = diag ? asum : 2.0 * asum;
}

// (block = 32 entries x 8 slices of the partial index, combined in a fixed order: the serial walk over N / 64
// partials took 21 us at N = 4096 and 48 us at N = 8192 on N / 256 workgroups)
template <typename R>
__global__ __launch_bounds__(256) void adjoint_u_kernel(const double* __restrict__ upart, const double* __restrict__ vpart,
                                                        int nt64, const R* __restrict__ q, const R* __restrict__ wl, int n,
                                                        int np, R* __restrict__ tvec, double* __restrict__ uq) {
  __shared__ double sl[8][33];
  const int li = threadIdx.x & 31, z = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + li;
  double u = 0.0;
  if (i < n) {
    const int T = i >> 6;
#pragma unroll 4
    for (int t = z; t < nt64; t += 8) u += (t <= T ? upart : vpart)[(int64_t)t * np + i];
  }
  sl[z][li] = u;
  __syncthreads();
  if (z != 0 || i >= np) return;
  if (i >= n) {
    tvec[i] = (R)0;
    uq[i] = 0.0;
    return;
  }
  u = sl[0][li];
#pragma unroll
  for (int k = 1; k < 8; ++k) u += sl[k][li];
  const double v = u / (double)q[i];
  uq[i] = v;
  tvec[i] = (R)(v - (double)wl[i]);
}

// the three sums behind the adjoint pass in one launch (block 0: uq, block 1: wl, block 2: the tile sums)
template <typename R>
__global__ void adjoint_sums_kernel(const double* __restrict__ uq, const R* __restrict__ wl, int n,
                                    const double* __restrict__ sumA_part, int ntile_tri, double* __restrict__ out) {
  __shared__ double sh[17];
  const int b = blockIdx.x;
  double v = 0.0;
  if (b == 0) for (int i = threadIdx.x; i < n; i += blockDim.x) v += uq[i];
  else if (b == 1) for (int i = threadIdx.x; i < n; i += blockDim.x) v += (double)wl[i];
  else for (int i = threadIdx.x; i < ntile_tri; i += blockDim.x) v += sumA_part[i];
  v = block_sum(v, sh);
  if (threadIdx.x == 0) out[b] = v;
}

// Rectangular adjoint (x1 != x2, utils.py:996-1021 contracted with W[n1][n2]): per 64 x 64 tile
//   Aout = W o (pi - delta)/pi ,  B_m = W o sqrt(1 - c^2)/pi ,
//   upart[tj][i] = sum_{j in tile} B_m[i][j] q2[j] ,  vpart[ti][j] = sum_{i in tile} B_m[i][j] q1[i] ,
//   tile_sum[ti][tj] = sum Aout.   Padding rows / columns of Aout are written as zero.
__global__ __launch_bounds__(256) void adjoint_rect_kernel(const double* __restrict__ W, int64_t ldw,
                                                           const double* __restrict__ Cos, int64_t ldc,
                                                           const double* __restrict__ q1,
                                                           const double* __restrict__ q2, int n1, int n2, int np1,
                                                           int np2, double* __restrict__ Aout, int64_t lda,
                                                           double* __restrict__ upart, double* __restrict__ vpart,
                                                           double* __restrict__ tile_sum) {
  __shared__ double colsum[4][64];
  __shared__ double sh[17];
  const int tj = blockIdx.x, ti = blockIdx.y;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = tj * 64 + tx;
  const double qj = (j < n2) ? q2[j] : 0.0;
  double csum = 0.0, asum = 0.0;
  for (int rr = ty; rr < 64; rr += 4) {
    const int i = ti * 64 + rr;
    double aw = 0.0, bm = 0.0;
    if (i < n1 && j < n2) {
      const double w = W[(int64_t)i * ldw + j];
      const double c = Cos[(int64_t)i * ldc + j];
      aw = w * (PI32 - acos(c)) / PI32;
      bm = w * sqrt(1.0 - c * c) / PI32;
    }
    Aout[(int64_t)i * lda + j] = aw;
    asum += aw;
    const double rs = wave_sum(bm * qj);
    if (tx == 0) upart[(int64_t)tj * np1 + i] = rs;
    csum += (i < n1) ? bm * q1[i] : 0.0;
  }
  colsum[ty][tx] = csum;
  __syncthreads();
  if (ty == 0) vpart[(int64_t)ti * np2 + j] = colsum[0][tx] + colsum[1][tx] + colsum[2][tx] + colsum[3][tx];
  asum = block_sum(asum, sh);
  if (threadIdx.x == 0) tile_sum[ti * gridDim.x + tj] = asum;
}

// u[i] = sum_t part[t][i];  tvec = u / (2 q) (+ extra);  uq = u / q   (zero on padding).
// Block = 32 entries x 8 slices of the partial index (fixed order): with one thread walking all nt = N / 64 partials
// the pass over n_t = 8192 tile rows took 44 us on 8 workgroups.
__global__ __launch_bounds__(256) void adjoint_rect_reduce_kernel(const double* __restrict__ part, int nt,
                                                                  const double* __restrict__ q,
                                                                  const double* __restrict__ extra, int n, int np,
                                                                  double* __restrict__ tvec, double* __restrict__ uq) {
  __shared__ double sl[8][33];
  const int li = threadIdx.x & 31, z = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + li;
  double u = 0.0;
  if (i < n) {
#pragma unroll 4
    for (int t = z; t < nt; t += 8) u += part[(int64_t)t * np + i];
  }
  sl[z][li] = u;
  __syncthreads();
  if (z != 0 || i >= np) return;
  if (i >= n) {
    tvec[i] = 0.0;
    uq[i] = 0.0;
    return;
  }
  u = sl[0][li];
#pragma unroll
  for (int k = 1; k < 8; ++k) u += sl[k][li];
  const double v = u / q[i];
  uq[i] = v;
  tvec[i] = 0.5 * v + (extra ? extra[i] : 0.0);
}

// up to four independent sums in one launch (block b: out[b] = scale[b] * sum of x[b][0 .. n[b]))
struct SumList {
  const double* x[4];
  int n[4];
  double scale[4];
  double* out[4];
};
__global__ void sum_list_kernel(SumList l) {
  __shared__ double sh[17];
  const int b = blockIdx.x;
  double v = 0.0;
  for (int i = threadIdx.x; i < l.n[b]; i += blockDim.x) v += l.x[b][i];
  v = block_sum(v, sh);
  if (threadIdx.x == 0) l.out[b][0] = l.scale[b] * v;
}

int launch_adjoint_rect(const double* W, int64_t ldw, const double* Cos, int64_t ldc, const double* q1,
                        const double* q2, int n1, int n2, int np1, int np2, double* Aout, int64_t lda, double* upart,
                        double* vpart, double* tile_sum, const double* extra1, double* t1, double* t2, double* uq1,
                        double* uq2, double* scal3, hipStream_t s) {
  const int g1 = np1 / 64, g2 = np2 / 64;
  hipLaunchKernelGGL(adjoint_rect_kernel, dim3(g2, g1), dim3(256), 0, s, W, ldw, Cos, ldc, q1, q2, n1, n2, np1, np2,
                     Aout, lda, upart, vpart, tile_sum);
  hipLaunchKernelGGL(adjoint_rect_reduce_kernel, dim3((np1 + 31) / 32), dim3(256), 0, s, upart, g2, q1, extra1, n1,
                     np1, t1, uq1);
  hipLaunchKernelGGL(adjoint_rect_reduce_kernel, dim3((np2 + 31) / 32), dim3(256), 0, s, vpart, g1, q2,
                     (const double*)nullptr, n2, np2, t2, uq2);
  SumList l{};
  l.x[0] = tile_sum; l.n[0] = g1 * g2; l.scale[0] = 1.0; l.out[0] = scal3 + 0;
  l.x[1] = uq1; l.n[1] = n1; l.scale[1] = 1.0; l.out[1] = scal3 + 1;
  l.x[2] = uq2; l.n[2] = n2; l.scale[2] = 1.0; l.out[2] = scal3 + 2;
  hipLaunchKernelGGL(sum_list_kernel, dim3(3), dim3(1024), 0, s, l);
  GP_HIP(hipGetLastError());
  return 0;
}

template <typename R>
int launch_adjoint(const R* W, const R* Cos, int64_t ld, const R* b, const R* q, int n, int np, R* Aout,
                   double* upart, double* vpart, double* sumA_part, hipStream_t s) {
  const int t = np / 64, nt = t * (t + 1) / 2;
  hipLaunchKernelGGL(adjoint_kernel<R>, dim3(nt), dim3(256), 0, s, W, Cos, ld, b, q, n, np, Aout, upart, vpart,
                     sumA_part);
  GP_HIP(hipGetLastError());
  return 0;
}

template <typename R>
int launch_adjoint_reduce(const double* upart, const double* vpart, const double* sumA_part, int ntile,
                          int ntile_tri, const R* q, const R* wl, int n, int np, R* tvec, double* uq,
                          double* scal_out, hipStream_t s) {
  hipLaunchKernelGGL(adjoint_u_kernel<R>, dim3((np + 31) / 32), dim3(256), 0, s, upart, vpart, ntile, q, wl, n,
                     np, tvec, uq);
  hipLaunchKernelGGL(adjoint_sums_kernel<R>, dim3(3), dim3(1024), 0, s, uq, wl, n, sumA_part, ntile_tri, scal_out);
  GP_HIP(hipGetLastError());
  return 0;
}

template <typename R>
__global__ void rowscale_add_kernel(R* __restrict__ Y, int64_t ldy, const R* __restrict__ Xm, int64_t ldm,
                                    const R* __restrict__ t, int dp) {
  const int i = blockIdx.y;
  const R ti = t[i];
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < dp; k += gridDim.x * blockDim.x)
    Y[(int64_t)i * ldy + k] += ti * Xm[(int64_t)i * ldm + k];
}

template <typename R>
int launch_rowscale_add(R* Y, int64_t ldy, const R* Xm, int64_t ldm, const R* t, int np, int dp, hipStream_t s) {
  hipLaunchKernelGGL(rowscale_add_kernel<R>, dim3((dp + 255) / 256, np), dim3(256), 0, s, Y, ldy, Xm, ldm, t, dp);
  GP_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ metric contraction
// grad5[p] = sum_kl dC_p[k][l] M[k][l], dC_p rebuilt from C and the pixel coordinates
// (utils.py:902-909), order Amp, -2log2beta, -log2rho2, eps_0x, eps_0y.
// METRIC_BLOCKS workgroups take the rows of the d x d matrices cyclically; each leaves its five partial
// sums in part[block][5] and the last one to arrive (ticket counter, reset for the next call) adds the
// partials in block order, so the result does not depend on the arrival order.  (The single-workgroup
// version of round 1 took 105 us at d = 256, at the end of every evaluation's critical path.)
constexpr int METRIC_BLOCKS = 32;
template <typename R>
__global__ __launch_bounds__(256) void metric_contract_kernel(Theta th, const int* __restrict__ pix, int d, int n_rows,
                                                              int n_cols, const R* __restrict__ C, int64_t ldc,
                                                              const R* __restrict__ M, int64_t ldm,
                                                              double* __restrict__ grad5, double* __restrict__ part,
                                                              int* __restrict__ ticket) {
  __shared__ double sh[17];
  __shared__ int last;
  double g[5] = {0, 0, 0, 0, 0};
  for (int i = blockIdx.x; i < d; i += gridDim.x) {
    const int pi = pix[i];
    const double xi = lin_pm1(pi % n_cols, n_cols), yi = lin_pm1(pi / n_cols, n_rows);
    const double dxi = xi - th.eps0x, dyi = yi - th.eps0y;
    const double lai = -th.eb * (dxi * dxi + dyi * dyi);
    for (int j = threadIdx.x; j < d; j += blockDim.x) {
      const int pj = pix[j];
      const double xj = lin_pm1(pj % n_cols, n_cols), yj = lin_pm1(pj / n_cols, n_rows);
      const double dxj = xj - th.eps0x, dyj = yj - th.eps0y;
      const double laj = -th.eb * (dxj * dxj + dyj * dyj);
      const double ex = xj - xi, ey = yj - yi;
      const double ls = -th.er * (ex * ex + ey * ey);
      const double c = (double)C[(int64_t)i * ldc + j];
      const double mm = (double)M[(int64_t)i * ldm + j];
      g[0] += (c / th.amp) * mm;
      g[1] += (c * (lai + laj)) * mm;
      g[2] += (c * ls) * mm;
      g[3] += (2.0 * th.eb * c * (xi + xj - 2.0 * th.eps0x)) * mm;
      g[4] += (2.0 * th.eb * c * (yi + yj - 2.0 * th.eps0y)) * mm;
    }
  }
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    const double v = block_sum(g[p], sh);
    if (threadIdx.x == 0) part[blockIdx.x * 5 + p] = v;
  }
  if (threadIdx.x == 0) {
    __threadfence();                                    // partials visible before the ticket
    last = (atomicAdd(ticket, 1) == (int)gridDim.x - 1);
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (threadIdx.x < 5) {
    double v = 0.0;
    for (int b = 0; b < (int)gridDim.x; ++b) v += __hip_atomic_load(&part[b * 5 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    grad5[threadIdx.x] = v;
  }
  if (threadIdx.x == 0) *ticket = 0;
}

// part: >= 5 * METRIC_BLOCKS doubles of scratch; ticket: a device int that is 0 between calls
template <typename R>
int launch_metric_contract(const Theta& th, const int* pix, int d, int n_rows, int n_cols, const R* C, int64_t ldc,
                           const R* M, int64_t ldm, double* grad5, double* part, int* ticket, hipStream_t s) {
  hipLaunchKernelGGL(metric_contract_kernel<R>, dim3(METRIC_BLOCKS), dim3(256), 0, s, th, pix, d, n_rows, n_cols, C,
                     ldc, M, ldm, grad5, part, ticket);
  GP_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ generic API helpers
// dst[dp][ldd] <- src[d][lds] zero padded
__global__ void pad_copy_kernel(const double* __restrict__ src, int64_t lds, int rows, int cols,
                                double* __restrict__ dst, int64_t ldd, int prow, int pcol) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= pcol || i >= prow) return;
  dst[(int64_t)i * ldd + j] = (i < rows && j < cols) ? src[(int64_t)i * lds + j] : 0.0;
}

int launch_pad_copy(const double* src, int64_t lds, int rows, int cols, double* dst, int64_t ldd, int prow,
                    int pcol, hipStream_t s) {
  hipLaunchKernelGGL(pad_copy_kernel, dim3((pcol + 255) / 256, prow), dim3(256), 0, s, src, lds, rows, cols, dst,
                     ldd, prow, pcol);
  GP_HIP(hipGetLastError());
  return 0;
}

// K <- (K + K^T)/2 for a square n x n matrix (reference utils.py:1024-1025 when n1 == n2 but x1 != x2)
__global__ void symmetrize_avg_kernel(double* __restrict__ A, int64_t lda, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (i >= n || j >= i) return;
  const double a = A[(int64_t)i * lda + j], b = A[(int64_t)j * lda + i];
  const double v = (a + b) / 2.0;
  A[(int64_t)i * lda + j] = v;
  A[(int64_t)j * lda + i] = v;
}

int launch_symmetrize_avg(double* A, int64_t lda, int n, hipStream_t s) {
  hipLaunchKernelGGL(symmetrize_avg_kernel, dim3((n + 255) / 256, n), dim3(256), 0, s, A, lda, n);
  GP_HIP(hipGetLastError());
  return 0;
}

// dK/dsigma_0 (utils.py:996-1004): pure element-wise from q1, q2 and cos(delta)
__global__ void dk_sigma0_kernel(const double* __restrict__ Cos, int64_t ldc, const double* __restrict__ q1,
                                 const double* __restrict__ q2, int n1, int n2, double s0,
                                 double* __restrict__ dK, int64_t ldk) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (i >= n1 || j >= n2) return;
  const double c = Cos[(int64_t)i * ldc + j];
  const double a = q1[i], b = q2[j], qq = a * b;
  const double delta = acos(c);
  const double J = (sqrt(1.0 - c * c) + PI32 * c - delta * c) / PI32;
  const double dqq = s0 * s0 * (b / a + a / b);        // :996
  const double dcos = (2.0 * s0 * s0 - c * dqq) / qq;  // :998
  const double dJ = -(delta - PI32) * dcos / PI32;     // :1000
  dK[(int64_t)i * ldk + j] = (qq * dJ + dqq * J) / s0; // :1004
}

int launch_dk_sigma0(const double* Cos, int64_t ldc, const double* q1, const double* q2, int n1, int n2,
                     double s0, double* dK, int64_t ldk, hipStream_t s) {
  hipLaunchKernelGGL(dk_sigma0_kernel, dim3((n2 + 255) / 256, n1), dim3(256), 0, s, Cos, ldc, q1, q2, n1, n2, s0,
                     dK, ldk);
  GP_HIP(hipGetLastError());
  return 0;
}

// dq[i] = 0.5 * sum_k Xt[k][i] * XDt[k][i] / q[i]   (utils.py:1012-1013);  h = the plain sum (:1042)
__global__ void dq_kernel(const double* __restrict__ Xt, const double* __restrict__ XDt, int64_t ld, int dp,
                          int n, const double* __restrict__ q, double* __restrict__ dq, double* __restrict__ h) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = 0.0;
  for (int k = 0; k < dp; ++k) v += Xt[(int64_t)k * ld + i] * XDt[(int64_t)k * ld + i];
  if (h) h[i] = v;
  if (dq) dq[i] = 0.5 * v / q[i];
}

int launch_dq(const double* Xt, const double* XDt, int64_t ld, int dp, int n, const double* q, double* dq,
              double* h, hipStream_t s) {
  hipLaunchKernelGGL(dq_kernel, dim3((n + 255) / 256), dim3(256), 0, s, Xt, XDt, ld, dp, n, q, dq, h);
  GP_HIP(hipGetLastError());
  return 0;
}

// dK_p = qq dJ + dqq J from H = x1 dC_p x2^T (utils.py:1015-1021); H is overwritten in place
__global__ void dk_metric_kernel(double* __restrict__ H, int64_t ldh, const double* __restrict__ Cos, int64_t ldc,
                                 const double* __restrict__ q1, const double* __restrict__ q2,
                                 const double* __restrict__ dq1, const double* __restrict__ dq2, int n1, int n2) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (i >= n1 || j >= n2) return;
  const double c = Cos[(int64_t)i * ldc + j];
  const double a = q1[i], b = q2[j], qq = a * b;
  const double delta = acos(c);
  const double J = (sqrt(1.0 - c * c) + PI32 * c - delta * c) / PI32;
  const double dqq = dq1[i] * b + a * dq2[j];                       // :1015
  const double dcos = (H[(int64_t)i * ldh + j] - c * dqq) / qq;      // :1017
  const double dJ = -(delta - PI32) * dcos / PI32;                   // :1019
  H[(int64_t)i * ldh + j] = qq * dJ + dqq * J;                       // :1021
}

int launch_dk_metric(double* H, int64_t ldh, const double* Cos, int64_t ldc, const double* q1, const double* q2,
                     const double* dq1, const double* dq2, int n1, int n2, hipStream_t s) {
  hipLaunchKernelGGL(dk_metric_kernel, dim3((n2 + 255) / 256, n1), dim3(256), 0, s, H, ldh, Cos, ldc, q1, q2, dq1,
                     dq2, n1, n2);
  GP_HIP(hipGetLastError());
  return 0;
}

template <typename R>
__global__ void add_diag_kernel(R* __restrict__ A, int64_t lda, int n, double v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) A[(int64_t)i * lda + i] += (R)v;
}

template <typename R>
int launch_add_diag(R* A, int64_t lda, int n, double v, hipStream_t s) {
  hipLaunchKernelGGL(add_diag_kernel<R>, dim3((n + 255) / 256), dim3(256), 0, s, A, lda, n, v);
  GP_HIP(hipGetLastError());
  return 0;
}

// dst[i] = alpha * src[i]
template <typename R>
__global__ void scale_copy_kernel(R* __restrict__ dst, const R* __restrict__ src, int n, R alpha) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = alpha * src[i];
}

template <typename R>
int launch_scale_copy(R* dst, const R* src, int n, double alpha, hipStream_t s) {
  hipLaunchKernelGGL(scale_copy_kernel<R>, dim3((n + 255) / 256), dim3(256), 0, s, dst, src, n, (R)alpha);
  GP_HIP(hipGetLastError());
  return 0;
}

// dst[r][c] = a * dst[r][c] + b * src[r][c] over a rows x cols block (cols even, 16-byte rows)
template <typename R>
__global__ void axpby_block_kernel(R* __restrict__ dst, int64_t ldd, const R* __restrict__ src, int64_t lds, int rows,
                                   int cols, R a, R b) {
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
  const int r = blockIdx.y;
  if (c >= cols || r >= rows) return;
  R* d = dst + (int64_t)r * ldd + c;
  const R* q = src + (int64_t)r * lds + c;
  d[0] = a * d[0] + b * q[0];
  d[1] = a * d[1] + b * q[1];
}

template <typename R>
int launch_axpby_block(R* dst, int64_t ldd, const R* src, int64_t lds, int rows, int cols, double a, double b,
                       hipStream_t s) {
  hipLaunchKernelGGL(axpby_block_kernel<R>, dim3((cols / 2 + 255) / 256, rows), dim3(256), 0, s, dst, ldd, src, lds,
                     rows, cols, (R)a, (R)b);
  GP_HIP(hipGetLastError());
  return 0;
}

__global__ void fill_kernel(double* __restrict__ x, int64_t n, double v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = v;
}

int launch_fill(double* x, int64_t n, double v, hipStream_t s) {
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, n, v);
  GP_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ E-step helpers
// s_i = A sqrt(f_i) ; rhs_i = A^2 f_i m_i + A (r_i - f_i)   (utils.py:1421-1422, 1431 with a = I)
__global__ void estep_prep_kernel(const double* __restrict__ f, const double* __restrict__ r,
                                  const double* __restrict__ m, int n, int np, double A, double* __restrict__ sv,
                                  double* __restrict__ rhs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
  if (i >= n) {
    sv[i] = 0.0;
    rhs[i] = 0.0;
    return;
  }
  sv[i] = A * sqrt(f[i]);
  rhs[i] = A * A * f[i] * m[i] + A * (r[i] - f[i]);
}

int launch_estep_prep(const double* f, const double* r, const double* m, int n, int np, double A, double* sv,
                      double* rhs, hipStream_t s) {
  hipLaunchKernelGGL(estep_prep_kernel, dim3((np + 255) / 256), dim3(256), 0, s, f, r, m, n, np, A, sv, rhs);
  GP_HIP(hipGetLastError());
  return 0;
}

// M = I + diag(s) K diag(s) (lower 128-tiles, identity padding), SK = diag(s) K (full, zero padded),
// Kl = K (lower 128-tiles, zero padded) -- one pass over K
__global__ void estep_build_kernel(const double* __restrict__ K, int64_t ldk, int n, const double* __restrict__ sv,
                                   double* __restrict__ Mb, double* __restrict__ SK, double* __restrict__ Kl,
                                   int64_t ld) {
  const int tj = blockIdx.x, ti = blockIdx.y;
  const int r0 = ti * TILE, c0 = tj * TILE;
  for (int e = threadIdx.x; e < TILE * TILE; e += blockDim.x) {
    const int i = r0 + (e >> 7), j = c0 + (e & 127);
    const bool in = (i < n && j < n);
    const double k = in ? K[(int64_t)i * ldk + j] : 0.0;
    const double si = sv[i];
    SK[(int64_t)i * ld + j] = si * k;
    if (tj <= ti) {
      Mb[(int64_t)i * ld + j] = si * k * sv[j] + ((i == j) ? 1.0 : 0.0);
      Kl[(int64_t)i * ld + j] = k;
    }
  }
}

int launch_estep_build(const double* K, int64_t ldk, int n, int np, const double* sv, double* Mb, double* SK,
                       double* Kl, int64_t ld, hipStream_t s) {
  hipLaunchKernelGGL(estep_build_kernel, dim3(np / TILE, np / TILE), dim3(256), 0, s, K, ldk, n, sv, Mb, SK, Kl, ld);
  GP_HIP(hipGetLastError());
  return 0;
}

// y = A x for a symmetric matrix stored in its lower triangle (row-major): one wave per row
// gathers the row part (j <= i) and the column part (j > i) of row i.
__global__ void symv_lower_kernel(const double* __restrict__ A, int64_t lda, int n, const double* __restrict__ x,
                                  double* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= n) return;
  double v = 0.0;
  for (int j = lane; j <= i; j += 64) v += A[(int64_t)i * lda + j] * x[j];
  for (int j = i + 1 + lane; j < n; j += 64) v += A[(int64_t)j * lda + i] * x[j];
  v = wave_sum(v);
  if (lane == 0) y[i] = v;
}

int launch_symv_lower(const double* A, int64_t lda, int n, const double* x, double* y, hipStream_t s) {
  hipLaunchKernelGGL(symv_lower_kernel, dim3((n + 3) / 4), dim3(256), 0, s, A, lda, n, x, y);
  GP_HIP(hipGetLastError());
  return 0;
}

// dst[n][ldd] (full symmetric) <- lower triangle of src[np][lds]
__global__ void unpack_sym_kernel(const double* __restrict__ src, int64_t lds, int n, double* __restrict__ dst,
                                  int64_t ldd) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (i >= n || j >= n) return;
  dst[(int64_t)i * ldd + j] = (j <= i) ? src[(int64_t)i * lds + j] : src[(int64_t)j * lds + i];
}

int launch_unpack_sym(const double* src, int64_t lds, int n, double* dst, int64_t ldd, hipStream_t s) {
  hipLaunchKernelGGL(unpack_sym_kernel, dim3((n + 255) / 256, n), dim3(256), 0, s, src, lds, n, dst, ldd);
  GP_HIP(hipGetLastError());
  return 0;
}

// dst[n][ldd] <- lower triangle of src, strict upper zeroed (triangular factor for the caller)
__global__ void unpack_tri_kernel(const double* __restrict__ src, int64_t lds, int n, double* __restrict__ dst,
                                  int64_t ldd) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (i >= n || j >= n) return;
  dst[(int64_t)i * ldd + j] = (j <= i) ? src[(int64_t)i * lds + j] : 0.0;
}

int launch_unpack_tri(const double* src, int64_t lds, int n, double* dst, int64_t ldd, hipStream_t s) {
  hipLaunchKernelGGL(unpack_tri_kernel, dim3((n + 255) / 256, n), dim3(256), 0, s, src, lds, n, dst, ldd);
  GP_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ firing-rate parameters
// One pass over the N training points for the inner logA optimiser (utils.py:1897-1934):
//   e_i = exp(A lam_m_i + A^2/2 lam_var_i)
//   lambda0 = closed form log(sum r) - log(sum e)            (utils.py:1215-1229) or the given one
//   f_i = e_i exp(lambda0)                                   (utils.py:1138)
//   loglik = A r.lam_m + lambda0 sum r - sum f               (utils.py:1243)
//   dloglik/dlogA = A (r.lam_m - sum (lam_m + A lam_var) f)  (utils.py:1253)
// out[0] lambda0 used, out[1] loglik, out[2] dloglik/dlogA, out[3] sum f, out[4] sum r, out[5] r.lam_m,
// out[6] closed-form lambda0
__global__ void fparam_kernel(const double* __restrict__ lam_m, const double* __restrict__ lam_var,
                              const double* __restrict__ r, int n, double A, int closed_form, double lambda0_in,
                              double* __restrict__ f, double* __restrict__ out) {
  __shared__ double sh[17];
  double se = 0.0, sr = 0.0, srm = 0.0, sg = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double e = exp(A * lam_m[i] + 0.5 * A * A * lam_var[i]);
    se += e;
    sr += r[i];
    srm += r[i] * lam_m[i];
    sg += (lam_m[i] + A * lam_var[i]) * e;
  }
  se = block_sum(se, sh);
  sr = block_sum(sr, sh);
  srm = block_sum(srm, sh);
  sg = block_sum(sg, sh);
  const double lambda0 = closed_form ? (log(sr) - log(se)) : lambda0_in;
  const double el0 = exp(lambda0);
  if (f)
    for (int i = threadIdx.x; i < n; i += blockDim.x)
      f[i] = exp(A * lam_m[i] + 0.5 * A * A * lam_var[i] + lambda0);
  if (threadIdx.x == 0) {
    const double sf = se * el0;
    out[0] = lambda0;
    out[1] = A * srm + lambda0 * sr - sf;
    out[2] = A * (srm - sg * el0);
    out[3] = sf;
    out[4] = sr;
    out[5] = srm;
    out[6] = log(sr) - log(se);  // closed-form lambda0 for this logA, whatever was used above
  }
}

int launch_fparam(const double* lam_m, const double* lam_var, const double* r, int n, double A, int closed_form,
                  double lambda0_in, double* f, double* out, hipStream_t s) {
  hipLaunchKernelGGL(fparam_kernel, dim3(1), dim3(1024), 0, s, lam_m, lam_var, r, n, A, closed_form, lambda0_in, f,
                     out);
  GP_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ explicit instantiations
#define GP_INST(R)                                                                                                  \
  template int launch_localker<R>(const Theta&, const int*, int, int, int, int, R*, int64_t, R*, hipStream_t);      \
  template int launch_gather<R>(const R*, int64_t, int, const int*, int, int, int, R*, int64_t, R*, int64_t,        \
                                hipStream_t);                                                                       \
  template int launch_qvec<R>(const R*, const R*, int64_t, int, int, int, double, R*, R*, hipStream_t);             \
  template int launch_pack_lower<R>(const R*, int64_t, int, R*, int64_t, int, hipStream_t);                         \
  template int launch_symmetrize<R>(R*, int64_t, int, hipStream_t);                                                 \
  template int launch_logdet<R>(const R*, int64_t, int, double*, hipStream_t);                                      \
  template int launch_logdet_pair<R>(const R*, double*, const R*, double*, int64_t, int, hipStream_t);              \
  template int launch_frob_lower<R>(const R*, int64_t, int, double*, double*, hipStream_t);                         \
  template int launch_trmv_lower<R>(const R*, int64_t, int, const R*, R*, hipStream_t);                             \
  template int launch_trmv_lower_t<R>(const R*, int64_t, int, const R*, R*, double*, hipStream_t);                  \
  template int launch_group_prepare<R>(const GroupPrepT<R>&, hipStream_t);                                          \
  template int launch_dot<R>(const R*, const R*, int, double*, hipStream_t);                                        \
  template int launch_moments<R>(const R*, const R*, const R*, int64_t, const R*, int64_t, const R*, const R*, int, \
                                 double, double, R*, R*, R*, R*, double*, double*, int*, hipStream_t);              \
  template int launch_adjoint<R>(const R*, const R*, int64_t, const R*, const R*, int, int, R*, double*, double*,   \
                                 double*, hipStream_t);                                                             \
  template int launch_adjoint_reduce<R>(const double*, const double*, const double*, int, int, const R*, const R*,  \
                                        int, int, R*, double*, double*, hipStream_t);                               \
  template int launch_rowscale_add<R>(R*, int64_t, const R*, int64_t, const R*, int, int, hipStream_t);             \
  template int launch_reduce_slices<R, R>(const R*, int64_t, int, R*, int64_t, hipStream_t);                        \
  template int launch_metric_contract<R>(const Theta&, const int*, int, int, int, const R*, int64_t, const R*,      \
                                         int64_t, double*, double*, int*, hipStream_t);                             \
  template int launch_add_diag<R>(R*, int64_t, int, double, hipStream_t);                                           \
  template int launch_axpby_block<R>(R*, int64_t, const R*, int64_t, int, int, double, double, hipStream_t);          \
  template int launch_scale_copy<R>(R*, const R*, int, double, hipStream_t);
GP_INST(double)
GP_INST(float)
#undef GP_INST
template int launch_reduce_slices<double, float>(const double*, int64_t, int, float*, int64_t, hipStream_t);

}  // namespace gpfit
