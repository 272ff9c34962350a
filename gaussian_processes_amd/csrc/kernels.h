// Launchers of the non-GEMM kernels (elementwise / reductions / leaf).  Internal header.
#pragma once
#include "common.h"

namespace gpfit {

// Hyperparameters in the reference's dict order (utils.py:824) plus derived constants.
struct Theta {
  double sigma0, eps0x, eps0y, logbeta, logrho, amp;  // as given
  double eb, er;                                      // exp(-2log2beta), exp(-log2rho2)
};

// ---- chol_leaf.hip
template <typename R>
int launch_chol_leaf(const R* A, int64_t lda, R* L, int64_t ldl, R* Linv, int64_t ldi, int* info, int info_base,
                     hipStream_t s);

// ---- chol_leaf_reg.hip: same contract, matrix in registers, 21 KiB of LDS (co-resident with GEMM workgroups)
template <typename R>
int launch_chol_leaf_reg(const R* A, int64_t lda, R* L, int64_t ldl, R* Linv, int64_t ldi, int* info, int info_base,
                         hipStream_t s);
// n independent 128 x 128 blocks in one launch (one workgroup each): block b factors A[b] into L[b], Li[b] and
// reports into info[b]; common leading dimensions and info_base
// Per-unit housekeeping of a group of evaluations, one launch for the whole group instead of four small copies /
// fills per unit at the start and two device-to-host copies per unit at the end (each of those is a blit kernel with
// a host round trip behind it: 1.8 ms per group of 16 at the end alone, profiles/r03_group_idle.txt).
// Begin: the masked pixel list from its pinned host copy, info zeroed, the mean zero-padded to np.
// End: the unit's 64 scalars and 4 info words written straight into its pinned (device-visible) host buffers.
template <typename R>
struct GroupPrepT {
  int n_units, n, np;
  const int* pix_host[GEMM_MAXB / 2];
  int* pix[GEMM_MAXB / 2];
  int d[GEMM_MAXB / 2];
  int* info[GEMM_MAXB / 2];
  const R* m[GEMM_MAXB / 2];
  R* mpad[GEMM_MAXB / 2];
};
struct GroupCollectT {
  int n_units;
  const double* scal[GEMM_MAXB / 2];
  const int* info[GEMM_MAXB / 2];
  double* scal_host[GEMM_MAXB / 2];
  int* info_host[GEMM_MAXB / 2];
};
template <typename R>
int launch_group_prepare(const GroupPrepT<R>& g, hipStream_t s);
int launch_group_collect(const GroupCollectT& g, hipStream_t s);

template <typename R>
struct LeafBatchT {
  const R* A[GEMM_MAXB];
  R* L[GEMM_MAXB];
  R* Li[GEMM_MAXB];
  int* info[GEMM_MAXB];
  int64_t lda, ldl, ldi;
  int info_base;
  int n;
};
template <typename R> int launch_chol_leaf_batch(const LeafBatchT<R>& bt, hipStream_t s);

// ---- elementwise.hip  (templated on the scalar type R = double | float; reductions are always
//      accumulated and returned in fp64)
// Spatial metric C (utils.py:861-914) over the masked pixels pix[d] of an n_rows x n_cols
// grid.  C is written zero-padded to [dp][ldc]; dC (optional) as 5 dense [d][d] matrices in
// the order Amp, -2log2beta, -log2rho2, eps_0x, eps_0y (the reference's dict order, :910).
template <typename R>
int launch_localker(const Theta& th, const int* pix, int d, int dp, int n_rows, int n_cols, R* C, int64_t ldc,
                    R* dC, hipStream_t s);
// Xt[k][n] = X[n][pix[k]] (k-major, zero padded to [dp][ldt]) and optionally the row-major
// masked copy Xm[n][k] ([np][ldm], zero padded).
template <typename R>
int launch_gather(const R* X, int64_t ldx, int n, const int* pix, int d, int dp, int np, R* Xt, int64_t ldt, R* Xm,
                  int64_t ldm, hipStream_t s);
// h[n] = sum_k Xt[k][n]*XCt[k][n];  Kvec = h + s0^2;  q = sqrt(Kvec)  (q = 1 on padding)
template <typename R>
int launch_qvec(const R* Xt, const R* XCt, int64_t ld, int dp, int n, int np, double s0sq, R* Kvec, R* q,
                hipStream_t s);
// dst lower tiles <- src (n x n, ld lds) with identity padding up to np.
template <typename R>
int launch_pack_lower(const R* src, int64_t lds, int n, R* dst, int64_t ldd, int np, hipStream_t s);
// mirror the lower triangle of an n x n matrix into its upper triangle (in place)
template <typename R> int launch_symmetrize(R* A, int64_t lda, int n, hipStream_t s);
// out[0] = 2*sum_i log(L_ii), i < n
template <typename R> int launch_logdet(const R* L, int64_t ldl, int n, double* out, hipStream_t s);
template <typename R> int launch_logdet_pair(const R* L0, double* out0, const R* L1, double* out1, int64_t ldl, int n, hipStream_t s);
// out[0] = sum over the lower-triangular tiles of T^2 (T has exact zeros above its diagonal)
template <typename R>
int launch_frob_lower(const R* T, int64_t ldt, int np, double* out, double* partial, hipStream_t s);
// y = L x (L lower, row-major) ; and z = L^T x
template <typename R> int launch_trmv_lower(const R* L, int64_t ldl, int np, const R* x, R* y, hipStream_t s);
template <typename R>
int launch_trmv_lower_t(const R* L, int64_t ldl, int np, const R* x, R* z, double* partial, hipStream_t s);
// out[0] = x . y
template <typename R> int launch_dot(const R* x, const R* y, int n, double* out, hipStream_t s);

// Latent moments / rate / likelihood pieces in the full-rank original basis:
//   lam_m = m ; lam_var = Kvec - K~_ii + V_ii ; f = exp(A lam_m + A^2/2 lam_var + lambda0)
//   scal[0] = r.lam_m  scal[1] = sum r  scal[2] = sum f
//   wl_i = -1/2 A^2 f_i g_i  with g_i = 1 - J_ii - (pi - delta_ii)(1 - c_ii)/pi
template <typename R>
int launch_moments(const R* Kvec, const R* q, const R* Cos, int64_t ldc, const R* V, int64_t ldv, const R* m,
                   const R* r, int n, double A, double lambda0, R* lam_m, R* lam_var, R* f, R* wl, double* scal,
                   double* part, int* ticket, hipStream_t s);   // part: 3 ceil(n / 256) doubles; ticket: 0 between calls

// Adjoint pass over the lower tiles of W (np x np):
//   w = W_ij - 1/2 b_i b_j ; Aw = w (pi - acos c)/pi ; Bm = w sqrt(1-c^2)/pi
//   Aw written to Aout symmetric (both triangles), zero on padding;
//   upart[tj][i] = sum_{j in tile tj} Bm_ij q_j  (+ mirrored contribution -> vpart[ti][j])
//   sumA_part[tile] = sum of Aw over the tile (off-diagonal tiles counted twice)
template <typename R>
int launch_adjoint(const R* W, const R* Cos, int64_t ld, const R* b, const R* q, int n, int np, R* Aout,
                   double* upart, double* vpart, double* sumA_part, hipStream_t s);
// u_i = sum_t upart[t][i] + vpart[t][i];  tvec_i = u_i/q_i - wl_i (uq: fp64 scratch [np]);
// scal_out[0] = sum_i u_i/q_i, scal_out[1] = sum_i wl_i, scal_out[2] = sum of sumA_part
template <typename R>
int launch_adjoint_reduce(const double* upart, const double* vpart, const double* sumA_part, int ntile,
                          int ntile_tri, const R* q, const R* wl, int n, int np, R* tvec, double* uq,
                          double* scal_out, hipStream_t s);
// Y[n][k] += t[n] * Xm[n][k]
template <typename R>
int launch_rowscale_add(R* Y, int64_t ldy, const R* Xm, int64_t ldm, const R* t, int np, int dp, hipStream_t s);
// dst = sum_z src[z] (split-K reduction), count elements each
template <typename RI, typename RO>
int launch_reduce_slices(const RI* src, int64_t slice_stride, int nslice, RO* dst, int64_t count, hipStream_t s);
// grad5[p] = sum_kl dC_p[k][l] * M[k][l] for the five metric hyperparameters, dC recomputed from C
template <typename R>
int launch_metric_contract(const Theta& th, const int* pix, int d, int n_rows, int n_cols, const R* C, int64_t ldc,
                           const R* M, int64_t ldm, double* grad5, double* part /* >= 160 doubles */,
                           int* ticket /* device int, 0 between calls */, hipStream_t s);
int launch_frob_finish(const double* partial, int nt, double* out, hipStream_t s);
template <typename R> int launch_add_diag(R* A, int64_t lda, int n, double v, hipStream_t s);
template <typename R> int launch_scale_copy(R* dst, const R* src, int n, double alpha, hipStream_t s);
// dst = a * dst + b * src over a rows x cols block (cols even)
template <typename R>
int launch_axpby_block(R* dst, int64_t ldd, const R* src, int64_t lds, int rows, int cols, double a, double b,
                       hipStream_t s);

// rectangular adjoint pass (W[n1][n2] against the analytic dK of acosker(x1, x2)): A_w, the two
// u vectors as t = u/(2q) (+ extra1) and u/q, scal3 = {sum A_w, sum u1/q1, sum u2/q2}
int launch_adjoint_rect(const double* W, int64_t ldw, const double* Cos, int64_t ldc, const double* q1,
                        const double* q2, int n1, int n2, int np1, int np2, double* Aout, int64_t lda, double* upart,
                        double* vpart, double* tile_sum, const double* extra1, double* t1, double* t2, double* uq1,
                        double* uq2, double* scal3, hipStream_t s);

// active-learning utility of nstar candidates (utils.py:413-525), r = list of response counts
int launch_nd_utility(const double* sigma2, const double* mu, int64_t nstar, const double* r, int nr, double* U,
                      hipStream_t s);

// ---- helpers of the general (materialising) acosker / localker entry points
int launch_pad_copy(const double* src, int64_t lds, int rows, int cols, double* dst, int64_t ldd, int prow,
                    int pcol, hipStream_t s);
int launch_symmetrize_avg(double* A, int64_t lda, int n, hipStream_t s);
int launch_dk_sigma0(const double* Cos, int64_t ldc, const double* q1, const double* q2, int n1, int n2,
                     double s0, double* dK, int64_t ldk, hipStream_t s);
int launch_dq(const double* Xt, const double* XDt, int64_t ld, int dp, int n, const double* q, double* dq,
              double* h, hipStream_t s);
int launch_dk_metric(double* H, int64_t ldh, const double* Cos, int64_t ldc, const double* q1, const double* q2,
                     const double* dq1, const double* dq2, int n1, int n2, hipStream_t s);
int launch_fill(double* x, int64_t n, double v, hipStream_t s);

// ---- projected.hip: element-wise pieces of the fused truncated-rank closure (gpfit_fit_eval_projected)
int launch_proj_moments(const double* Bp, const double* Kb, const double* aV, int64_t ld, int nb, const double* mb,
                        const double* Kvec, const double* r, int n, double A, double lambda0, double* lam_m,
                        double* lam_var, double* f, double* gm, double* gv, double* part, double* out3, hipStream_t s);
int launch_proj_ga(const double* Kb, const double* aV, int64_t ld, int nb, int n, int np, const double* gm,
                   const double* gv, const double* mb, double* Ga, hipStream_t s);
int launch_proj_gkb(const double* Bp, int64_t ld, int nb, int n, int np, const double* gv, double* GaKi, hipStream_t s);
int launch_proj_gktb(const double* Ki, const double* P1, const double* P2, int64_t ld, int nb, const double* b, double* G,
                     hipStream_t s);
int launch_proj_trace(const double* A, int64_t lda, int n, double* out, hipStream_t s);
// fused E-step in the projected basis (gpfit_estep_projected): per-row scalars s = A sqrt(f), u = A^2 f (a m) + A (r - f)
// over nrows >= n rows (zero on the padding); then Y = diag(s) aL zero-padded to [nrows][ld] (nrows a multiple of
// 32) with the slice sums of aL^T u in part[nrows / 32][npc]
int launch_estep_proj_rows(const double* a, int64_t lda, int nb, const double* mb, const double* f, const double* r,
                           int n, int nrows, double A, double* sv, double* u, hipStream_t s);
int launch_estep_proj_scale(const double* aL, int64_t ldal, int nb, int n, int nrows, const double* sv, const double* u,
                            double* Y, double* aLp /* or nullptr */, int64_t ld, int npc, double* part, hipStream_t s);
// lam_m = Z z1, lam_var = kv0 + row norms^2 of Z (Z = aL L_W^-T: the moments of lambda behind the update)
int launch_estep_proj_moments(const double* Z, int64_t ld, int nb, const double* z1, const double* kv0, int n,
                              double* lam_m, double* lam_var, hipStream_t s);

// ---- E-step / factorisation / firing-rate helpers
int launch_estep_prep(const double* f, const double* r, const double* m, int n, int np, double A, double* sv,
                      double* rhs, hipStream_t s);
int launch_estep_build(const double* K, int64_t ldk, int n, int np, const double* sv, double* Mb, double* SK,
                       double* Kl, int64_t ld, hipStream_t s);
int launch_symv_lower(const double* A, int64_t lda, int n, const double* x, double* y, hipStream_t s);
int launch_unpack_sym(const double* src, int64_t lds, int n, double* dst, int64_t ldd, hipStream_t s);
int launch_unpack_tri(const double* src, int64_t lds, int n, double* dst, int64_t ldd, hipStream_t s);
int launch_fparam(const double* lam_m, const double* lam_var, const double* r, int n, double A, int closed_form,
                  double lambda0_in, double* f, double* out, hipStream_t s);

}  // namespace gpfit
