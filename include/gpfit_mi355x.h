/* gpfit_mi355x.h -- C ABI of the MI355X-native GP fit path (libgpfit_mi355x.so).
 *
 * Drop-in boundary for the hot path of Spatial_GP_repo/utils.py (the reference has no
 * FFI layer of its own; the Python module gaussian_processes_amd/utils.py binds these
 * entry points with ctypes and keeps the reference's function signatures).
 *
 * Conventions
 *   - every matrix/vector pointer is a DEVICE pointer to contiguous row-major fp64 unless
 *     the parameter is documented as host; the caller owns every buffer; the library
 *     owns only the workspace inside a gpfit_ctx;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls enqueue
 *     work and return unless documented as synchronising;
 *   - return value: 0 ok; > 0 LAPACK-style info (1-based index of the first
 *     non-positive pivot); < 0 argument / runtime error (-2 = hyperparameter outside
 *     its limits, -3 = bad argument, -100 = HIP runtime error), message from
 *     gpfit_last_error();
 *   - hyperparameter vectors are double[6] in the reference's dict order
 *     sigma_0, eps_0x, eps_0y, -2log2beta, -log2rho2, Amp  (utils.py:824).
 */
#ifndef GPFIT_MI355X_H
#define GPFIT_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPFIT_VERSION 100

int gpfit_version(void);
const char* gpfit_last_error(void);

/* Raw fp64 MFMA GEMM:  C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] + beta * C.
 * Replaces the torch `@` / torch.matmul call sites of the path (utils.py:978-982,
 * 1012-1017, 2047-2062, 1318-1333).  a_kmajor: 0 = A stored [M][K], 1 = A stored [K][M];
 * b_kmajor: 1 = B stored [K][N], 0 = B stored [N][K].  out_lower: compute only the 128-tiles
 * on/below the diagonal; a_tri/b_tri: 0 dense, 1 op() lower-, 2 op() upper-triangular.
 * K % 16 == 0; M, N, lda, ldb even.  Runs on the caller's current device.  Large launches (>= 384
 * output tiles of 128 x 128, K >= 1024) may use the stream-K schedule; its partial-tile workspace is
 * kept per (device, stream) for this context-free entry point, so calls on different streams are
 * independent (the context-based entry points own their workspaces). */
int gpfit_dgemm(void* stream, int a_kmajor, int b_kmajor, int M, int N, int K, double alpha,
                const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* C,
                int64_t ldc, int out_lower, int a_tri, int b_tri);
/* Same with the tuning knobs exposed: walk (bit 0: tile grid backwards, bit 1: column-major -- also
 * for the lower triangle of an out_lower launch --, bit 2: every tile walks k downwards, bit 3: XCD-aware
 * macro-tile schedule for launches of >= 1536 tiles, bit 4: one workgroup per CU)
 * and tile (0 = automatic, or 128 / 64 / 32). */
int gpfit_dgemm_ex(void* stream, int a_kmajor, int b_kmajor, int M, int N, int K, double alpha,
                   const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* C,
                   int64_t ldc, int out_lower, int a_tri, int b_tri, int walk, int tile);

/* ---- workspace context -------------------------------------------------------------
 * One context per (device, maximum problem size); owns ~13 N^2 + O(N d) doubles of HBM
 * workspace, one auxiliary HIP stream and two events.  A context is NOT re-entrant: calls
 * on the same context must be serialised by the caller (one context per host thread; the Python
 * module keeps one per (device, thread)).  Every context entry point selects the context's device
 * for the duration of the call and restores the caller's current device on return; while an
 * asynchronous evaluation is pending (gpfit_fit_eval bit 2) every other entry point on that
 * context returns -3 until gpfit_fit_eval_finish has collected it.
 * n_max: stimuli; d_max: masked pixels; d_full_max: pixels of the full image. */
typedef struct gpfit_ctx gpfit_ctx;
int gpfit_ctx_create(int device, int64_t n_max, int64_t d_max, int64_t d_full_max, gpfit_ctx** out);
void gpfit_ctx_destroy(gpfit_ctx* ctx);

/* Hyperparameter box check of localker (utils.py:865-867) / closure_hyperparams
 * (utils.py:2020-2028).  0 inside, -2 outside (message names the offender).  Host only. */
int gpfit_check_limits(const double* theta, const double* lower, const double* upper);

/* Pixel mask of localker (utils.py:880-883) for an n_rows x n_cols grid: mask_host[p] = 1
 * where alpha_p >= 0.001; *d_out = number of kept pixels.  Host only (n_rows*n_cols exps). */
int gpfit_localker_mask(const double* theta, int n_rows, int n_cols, uint8_t* mask_host, int64_t* d_out);

/* localker (utils.py:861-914): spatial metric C[d][d] over the kept pixels of an n_rows x
 * n_cols grid and, when dC is non-NULL, its five derivatives as dC[5][d][d] in the order
 * Amp, -2log2beta, -log2rho2, eps_0x, eps_0y (the key order of the reference's dC dict,
 * utils.py:910).  mask_host/d come from gpfit_localker_mask.  Synchronises `stream`. */
int gpfit_localker(gpfit_ctx* ctx, void* stream, const double* theta, int n_rows, int n_cols,
                   const uint8_t* mask_host, int64_t d, double* C, double* dC);

/* acosker, diag=False (utils.py:968-1025): K[n1][n2] from already-masked x1[n1][d],
 * x2[n2][d] and the metric C[d][d]; symmetrised when n1 == n2 (utils.py:1024).  When dK is
 * non-NULL it receives six [n1][n2] matrices in theta dict order (sigma_0, eps_0x, eps_0y,
 * -2log2beta, -log2rho2, Amp); dC (layout as gpfit_localker) may be NULL, in which case only
 * the sigma_0 derivative is written.  Pass x2 == x1 for the square self-kernel (lower-tile
 * fast path).  Enqueues on `stream`, does not synchronise. */
int gpfit_acosker(gpfit_ctx* ctx, void* stream, double sigma0, const double* x1, int64_t ld1, int64_t n1,
                  const double* x2, int64_t ld2, int64_t n2, int64_t d, const double* C, int64_t ldC,
                  const double* dC, double* K, int64_t ldk, double* dK);

/* Rectangular pull-back for acosker(x1, x2), x1 != x2 (the K[n_t][n_tilde] of the sparse regime):
 * for an adjoint W[n1][n2] of K, the contraction sum_ij W_ij dK_p,ij with the reference's analytic
 * derivatives (utils.py:996-1021) reduces to <dC_p, sym(M)> for the five metric parameters and to
 * sigma_0 (2 out[0] + out[1] + out[2]) for sigma_0, with
 *   A_w = W o (pi - delta)/pi,  B_m = W o sqrt(1 - c^2)/pi,  u1 = B_m q2,  u2 = B_m^T q1,
 *   M = x1^T A_w x2 + x1^T diag(u1/(2 q1) + t1_extra) x1 + x2^T diag(u2/(2 q2)) x2      (d x d),
 *   out_host[3] = { sum A_w, sum u1/q1, sum u2/q2 }.
 * x1[n1][ld1], x2[n2][ld2] already masked, C[d][ldC]; t1_extra[n1] may be NULL (it carries the
 * dKvec adjoint of the closure); M_out[d][ldm] device.  Nothing n1 x n2 x 6 is materialised.
 * Synchronises. */
int gpfit_acosker_pullback(gpfit_ctx* ctx, void* stream, double sigma0, const double* x1, int64_t ld1,
                           int64_t n1, const double* x2, int64_t ld2, int64_t n2, int64_t d, const double* C,
                           int64_t ldC, const double* W, int64_t ldw, const double* t1_extra, double* M_out,
                           int64_t ldm, double* out_host);

/* acosker, diag=True (utils.py:1027-1044): Kvec[n1] = x_i C x_i + sigma_0^2 and optionally
 * dKvec[6][n1] (theta dict order). */
int gpfit_acosker_diag(gpfit_ctx* ctx, void* stream, double sigma0, const double* x1, int64_t ld1,
                       int64_t n1, int64_t d, const double* C, int64_t ldC, const double* dC, double* Kvec,
                       double* dKvec);

/* The fused unit of work: ONE evaluation of the M-step closure (utils.py:2017-2112) in the
 * full-rank regime (n_tilde == n_t, all eigenvalues kept), original basis:
 *   localker -> acosker (K~, Kvec) -> Cholesky(K~), Cholesky(V) -> lambda moments, f,
 *   log-likelihood, KL -> the six analytic gradients.
 * X[N][ldx] is the UN-masked stimulus matrix (device), r, m [N], V[N][ldv] symmetric (device).
 * out_host[16] (HOST): 0 loss = -(loglik - KL), 1 loglik, 2 KL, 3..8 d loss/d theta (dict order),
 *   9 log|K~|, 10 log|V|, 11 tr(K~^-1 V), 12 m^T K~^-1 m, 13 masked pixel count,
 *   14 info(K~), 15 info(V).
 * want_grad: bit 0 = compute the gradients; bit 1 = V is unchanged since the previous call on
 * this context (constant during an M-step): reuse its Cholesky factor and log-det; bit 2 =
 * asynchronous: only enqueue on `stream` and return (out_host is not written); the caller collects
 * the result with gpfit_fit_eval_finish.  Independent units (other cells, other theta points) can
 * then be kept in flight on several contexts / streams, so that one unit's latency-bound
 * factorisation overlaps another's GEMMs (multi.py); bit 3 = mixed precision (fp64 entry point only):
 * kernel build, both Cholesky factorisations, both log-determinants, the likelihood and m^T K~^-1 m in fp64, the
 * N^3-heavy products (T, Q, W and the pull-back) in fp32 on single-precision copies of the factors -- T's norm
 * included, so the trace term tr(K~^-1 V) of the loss is fp32-derived (measured effect on the loss: 2e-9 relative
 * over the 512-point lattice at N = 8192, asserted <= 1e-7 in tests/test_gpu_fp32.py) -- the hyperparameter-grid configuration (BASELINE configs[4]) at the north star's 1e-5
 * bar on the log marginal likelihood, which the all-fp32 instance misses on some grid points.
 * lam_m/lam_var/f (device, [N]) may be NULL.  Synchronises `stream` before returning unless bit 2.
 * Returns 0; -2 when theta is outside [lower, upper] (out_host[0] = +inf and gradients
 * +inf, exactly what the reference closure hands to L-BFGS); > 0 LAPACK info. */
int gpfit_fit_eval(gpfit_ctx* ctx, void* stream, const double* theta, const double* lower,
                   const double* upper, int n_rows, int n_cols, const double* X, int64_t ldx, int64_t N,
                   const double* r, const double* m, const double* V, int64_t ldv, double logA,
                   double lambda0, int want_grad, double* out_host, double* lam_m, double* lam_var,
                   double* f);

/* Collect the evaluation enqueued on `ctx` by gpfit_fit_eval / gpfit_fit_eval_f32 with bit 2 of
 * want_grad: waits for its stream, writes out_host[16] and returns what the synchronous call
 * would have returned (0 or the LAPACK info).  -3 if nothing is pending.  A unit enqueued by
 * gpfit_fit_eval_batch waits for its GROUP's completion event instead of the stream, so a later group (on other
 * contexts) may already be running on the same stream while this one is collected. */
int gpfit_fit_eval_finish(gpfit_ctx* ctx, double* out_host);

/* n_units (1 .. 16) independent units of work of the same N in one call -- the cells / hyperparameter-grid
 * points of BASELINE configs[3], [4] (no dependency between units), unit u on its own context
 * ctxs[u] with theta[6 u ..], X[u], r[u], m[u], V[u] (device pointers; X may be the same matrix for every
 * unit), logA[u], lambda0[u].  Replaces n_units calls of the closure utils.py:2017-2112.  Everything runs on
 * `stream`: the 2 n_units Cholesky recursions in lock step (their latency-bound leaves and small products are
 * shared launches), the products behind them as pointer-batched launches wherever one unit's product cannot
 * fill the chip.  Each unit's results are bit-identical to gpfit_fit_eval on its own.  want_grad: bits 0, 1, 3
 * as gpfit_fit_eval; the call is always asynchronous: rc_out[u] = 0 -> collect unit u with
 * gpfit_fit_eval_finish(ctxs[u], ...); -2 -> theta outside the limits, out_host[16 u ..] already holds the
 * infinite loss / gradients and nothing is pending.  A group's per-unit housekeeping is two launches (pixel
 * lists / info words / padded means at the start; the 64 result scalars and info words of every unit written
 * straight into its pinned host buffers at the end, followed by the completion event gpfit_fit_eval_finish waits
 * for).  Returns 0 or < 0 (bad argument / capacity / HIP error). */
int gpfit_fit_eval_batch(gpfit_ctx* const* ctxs, int n_units, void* stream, const double* theta,
                         const double* lower, const double* upper, int n_rows, int n_cols,
                         const double* const* X, int64_t ldx, int64_t N, const double* const* r,
                         const double* const* m, const double* const* V, int64_t ldv, const double* logA,
                         const double* lambda0, int want_grad, double* out_host, int* rc_out);
/* The same for the fp32 instance of the library (gpfit_fit_eval_f32). */
int gpfit_fit_eval_batch_f32(gpfit_ctx* const* ctxs, int n_units, void* stream, const double* theta,
                             const double* lower, const double* upper, int n_rows, int n_cols,
                             const float* const* X, int64_t ldx, int64_t N, const float* const* r,
                             const float* const* m, const float* const* V, int64_t ldv, const double* logA,
                             const double* lambda0, int want_grad, double* out_host, int* rc_out);

/* Gradient pull-back for an externally supplied adjoint: out_host[6] (theta dict order) =
 *   sum_ij W_ij dK~_p,ij + sum_i gvec_i dKvec_p,i
 * with the reference's analytic derivatives dK~_p (acosker, utils.py:996-1021) and dKvec_p
 * (utils.py:1036-1044) at theta, without materialising any dK (the contraction is pulled back to
 * the d x d metric, as in gpfit_fit_eval).  X[N][ldx] un-masked stimuli, W[N][ldw] symmetric
 * (lower triangle read), gvec[N], all device.  Serves the truncated-rank (B-projected) closure
 * of utils.py:2047-2099, whose n x n adjoints lift to W = (B G_Kb~ + G_Kb) B^T.  Synchronises. */
int gpfit_grad_pullback(gpfit_ctx* ctx, void* stream, const double* theta, int n_rows, int n_cols,
                        const double* X, int64_t ldx, int64_t N, const double* W, int64_t ldw,
                        const double* gvec, double* out_host);

/* Truncated-rank M-step closure, fused: the reference's B-projected closure (utils.py:2030-2099) in the
 * regime its default EIGVAL_TOL produces at realistic sizes -- inducing set = training set, n_kept < N
 * eigen-directions kept.  X[N][ldx] un-masked stimuli, r[N], B[N][ldb] the kept eigenvectors (orthonormal
 * columns, utils.py:1685), m_b[n_kept], V_b[n_kept][ldvb], all device.  One call does the kernel build,
 * the projections K_b = K~ B and K~_b = sym(B^T K_b), the Cholesky factorisations of K~_b and V_b, the
 * moments / likelihood / KL of utils.py:1090-1326 with a = B (:2068), the adjoints of the loss with
 * respect to K_b, K~_b and Kvec (da_p of :1114 included), their lift to the N x N adjoint
 * W = (B G_K~b + G_Kb) B^T and its contraction with the analytic dK~_p / dKvec_p by the pull-back to
 * the metric -- no dK, no N x N x 6 tensor, nothing computed by the host framework.
 * out_host[16] as gpfit_fit_eval (9 log|K~_b|, 10 log|V_b|, 11 tr(K~_b^-1 V_b), 12 m_b^T K~_b^-1 m_b,
 * 14 / 15 LAPACK info of the two factorisations).  Returns 0; -2 outside the hyperparameter box
 * (infinite loss and gradients); > 0 when a factorisation meets a non-positive pivot (the caller then
 * takes the reference's eigen-fallback of log_det, utils.py:1279-1304).  Synchronises. */
int gpfit_fit_eval_projected(gpfit_ctx* ctx, void* stream, const double* theta, const double* lower,
                             const double* upper, int n_rows, int n_cols, const double* X, int64_t ldx, int64_t N,
                             const double* r, const double* B, int64_t ldb, int64_t n_kept, const double* m_b,
                             const double* V_b, int64_t ldvb, double logA, double lambda0, double* out_host);

/* Sparse M-step closure, fused: n_tilde < n_t inducing stimuli (the regime of the lab's own fits,
 * one_cell_fit.ipynb:89: n_t ~ 3160, n_tilde up to 2100): K[n_t][n_tilde] = acosker(x, xtilde) differs from
 * K~ = acosker(xtilde, xtilde), a = K_b K~_b^-1 (utils.py:1693, 2068) and da_p is non-zero (:1114).
 * X[N][ldx] training stimuli, Xtilde[Ntilde][ldxt] inducing stimuli (both un-masked), B[Ntilde][ldb] the
 * kept eigenvectors of K~, m_b[n_kept], V_b[n_kept][ldvb].  As gpfit_fit_eval_projected, with two
 * adjoints pulled back to the metric: B G_K~b B^T against dK~_p on the inducing stimuli and G_Kb B^T
 * against dK_p on (x, xtilde), the dKvec term riding on the latter.  Same out_host and return
 * conventions.  Synchronises. */
int gpfit_fit_eval_sparse(gpfit_ctx* ctx, void* stream, const double* theta, const double* lower, const double* upper,
                          int n_rows, int n_cols, const double* X, int64_t ldx, int64_t N, const double* Xtilde,
                          int64_t ldxt, int64_t Ntilde, const double* r, const double* B, int64_t ldb, int64_t n_kept,
                          const double* m_b, const double* V_b, int64_t ldvb, double logA, double lambda0,
                          double* out_host);

/* Cholesky factorisation A = L L^T of a symmetric positive definite n x n matrix (lower
 * triangle read) by the recursive MFMA algorithm; replaces torch.linalg.cholesky in log_det
 * (utils.py:1275) and, through L^-1, the LU torch.linalg.solve(., I) of the closure
 * (utils.py:2067).  L / Linv (device, may be NULL) receive the factor and its inverse with the
 * strict upper part zeroed; *logdet_host = 2 sum log L_ii (utils.py:1278).  Synchronises.
 * Returns 0, or the LAPACK info (1-based index of the first non-positive pivot). */
int gpfit_potrf(gpfit_ctx* ctx, void* stream, const double* A, int64_t lda, int64_t n, double* L, int64_t ldl,
                double* Linv, int64_t ldi, double* logdet_host, int* info_host);

/* Rank-1 append to a Cholesky factorisation: given L and L^-1 of the n x n matrix K (device, row-major,
 * leading dimensions > n so that row and column n exist) and kcol[0..n] = the new last column of
 * K' = [K k; k^T kappa] (device), writes row n of both factors in place,
 *   L'[n][:] = (l^T, lambda),  L'^-1[n][:] = (-(L^-T l)^T / lambda, 1 / lambda),  l = L^-1 k,
 *   lambda = sqrt(kappa - l.l),
 * zeroes column n above the diagonal and adds 2 log(lambda) to *logdet_inout_host.  O(n^2) instead of
 * the O(n^3) refactorisation: the closed loop of one_cell_active_training.ipynb appends one stimulus
 * per iteration and updates its kernel matrices 'by their latest column' (:1889-1891).  Synchronises.
 * Returns 0, or n + 1 (LAPACK info) when the Schur complement kappa - l.l is not positive. */
int gpfit_potrf_append(gpfit_ctx* ctx, void* stream, double* L, int64_t ldl, double* Linv, int64_t ldi, int64_t n,
                       const double* kcol, double* logdet_inout_host, int* info_host);

/* E-step Newton update of q(lambda~) = N(m, V) (Estep, alpha = 1 branch, utils.py:1420-1439) in
 * the original basis (a = I): with s = A sqrt(f), M = I + S K~ S = L_M L_M^T, T = L_M^-1 S K~:
 *   V_new = K~ - T^T T,  m_new = V_new (A^2 f o m + A (r - f)).
 * K~[N][ldk] symmetric (full), r, m, f [N] -> m_new [N], V_new[N][ldv] (full, symmetric).
 * Synchronises; returns LAPACK info if M is not positive definite. */
int gpfit_estep(gpfit_ctx* ctx, void* stream, const double* K, int64_t ldk, int64_t N, const double* r,
                const double* m, const double* f, double logA, double* m_new, double* V_new, int64_t ldv);

/* The same Newton update in the basis in force when K~ is truncated or the inducing set is a subset (the else branch of
 * varGP's E-step, utils.py:1880 -> Estep :1420-1439 with a = K K~^-1 in the B basis): with L = chol(K~_b) and
 * aL = a L supplied by the caller (both fixed between two kernel rebuilds, i.e. over the nEstep updates of an EM
 * iteration), s = A sqrt(f), Y = diag(s) aL:
 *   W = I + Y^T Y = I + L^T G L = L_W L_W^T        (G = A^2 a^T diag(f) a, :1422)
 *   V_new = P P^T, P = L L_W^-T                     (= solve(I + K~ G, K~), :1430)
 *   m_new = L W^-1 aL^T u,  u = A^2 f o (a m) + A (r - f)   (= V_new (G m + g), :1431)
 * a[N][lda], aL[N][ldal] (N x nb), L[nb][ldl] lower, r, f [N], m [nb] -> m_new [nb], V_new[nb][ldv] (full, symmetric).
 * Optionally (all three pointers or none) the moments of lambda behind the update, which varGP evaluates next
 * (lambda_moments, utils.py:1090, 1101): with kv0 = Kvec - rowsum(K o a) [N] (fixed between kernel rebuilds) and
 * Z = aL L_W^-T,  lam_m = a m_new = Z (L_W^-1 aL^T u),  lam_var = kv0 + diag(a V_new a^T) = kv0 + row norms^2 of Z.
 * One call, one synchronisation; returns LAPACK info if W is not positive definite. */
int gpfit_estep_projected(gpfit_ctx* ctx, void* stream, const double* a, int64_t lda, const double* aL, int64_t ldal,
                          const double* L, int64_t ldl, int64_t N, int64_t nb, const double* r, const double* m,
                          const double* f, double logA, double* m_new, double* V_new, int64_t ldv, const double* kv0,
                          double* lam_m_out, double* lam_var_out);

/* One pass over the training points for the firing-rate parameters (mean_f_given_lambda_moments
 * utils.py:1126-1141, lambda0_given_logA :1215-1229, compute_loglikelihood with
 * compute_grad_for_f_params :1243-1255).  out_host[7]: lambda0 used (closed form if requested,
 * else lambda0_in), loglik, d loglik / d logA, sum f, sum r, r.lam_m, closed-form lambda0.
 * f_out (device [N]) may be NULL. */
int gpfit_fparam_eval(gpfit_ctx* ctx, void* stream, const double* lam_m, const double* lam_var,
                      const double* r, int64_t N, double logA, int closed_form_lambda0, double lambda0_in,
                      double* f_out, double* out_host);

/* Active-learning utility of nstar candidate stimuli, U = H(r|x,D) - <H(r|f,x)>
 * (nd_utility with nd_p_r_given_xD, nd_lambda_r_mean, nd_mean_noise_entropy, utils.py:413-525;
 * call site one_cell_active_training.ipynb: u2d = nd_utility(logf_var, logf_mean, arange(100))).
 * sigma2, mu: device [nstar] (variance and mean of log f); r: device [nr] response counts;
 * U: device [nstar].  The principal-branch Lambert W (scipy on the host in the reference,
 * utils.py:464-466) is evaluated on the device.  Asynchronous on `stream`. */
int gpfit_nd_utility(void* stream, const double* sigma2, const double* mu, int64_t nstar, const double* r,
                     int nr, double* U);

/* Per-launch HIP-event timing of the dominant kernels during gpfit_fit_eval (bench.py's
 * roofline leg; adds two event records per launch, so leave it off when timing throughput).
 * out16: 0 sum of the 128-tile GEMM launch durations [ms] (the dominant kernel family, stream-K
 *        variant included), 1 flops those launches executed, 2 #launches, 3 sum of Cholesky-leaf
 *        durations [ms], 4 #leaves, 5 Gram kernel [ms], 6 its flops, 7 #, 8-10 the same three
 *        figures for the 64/32-tile instances, 11-12 duration [ms] and flops of the single largest
 *        GEMM launch (L^-1 L_V at the headline), 13-15 unused. */
int gpfit_set_profile(gpfit_ctx* ctx, int on);
int gpfit_get_profile(gpfit_ctx* ctx, double* out16);
/* Phase timing (gpfit_set_profile(ctx, 2)): eight HIP events per evaluation and none inside the
 * factorisations, so the timed run is the production schedule.  out8[i] = milliseconds from the start
 * of the last synchronous gpfit_fit_eval (with gradients) to: 0 start, 1 kernel build + moments done
 * (the two factorisation chains start here), 2 Cholesky + inverse of K~ done (main stream), 3 Cholesky
 * of V done (auxiliary stream), 4 T = L^-1 L_V and its norm done, 5 Q = I - T T^T done, 6 two-sided
 * product W done, 7 end (gradient contraction, scalars copied). */
int gpfit_get_phases(gpfit_ctx* ctx, double* out8);
/* Host milliseconds the last gpfit_fit_eval spent enqueuing work (before its final sync). */
double gpfit_last_enqueue_ms(gpfit_ctx* ctx);

/* fp32 instance of the fused unit of work (hyperparameter-grid configuration, BASELINE
 * configs[4]; the reference itself is fp64-only, utils.py:31-33): X, r, m, V and the optional
 * output vectors are float32 device buffers, every factorisation / GEMM runs in fp32 on
 * v_mfma_f32_16x16x4_f32, reductions and the returned scalars are fp64.  Same contract
 * otherwise.  Accuracy against the fp64 path is stated in tests/test_gpu_fp32.py. */
int gpfit_fit_eval_f32(gpfit_ctx* ctx, void* stream, const double* theta, const double* lower,
                       const double* upper, int n_rows, int n_cols, const float* X, int64_t ldx, int64_t N,
                       const float* r, const float* m, const float* V, int64_t ldv, double logA, double lambda0,
                       int want_grad, double* out_host, float* lam_m, float* lam_var, float* f);

/* Roofline probes (no reference counterpart): back-to-back v_mfma_f64_16x16x4_f64 issue
 * (flops = blocks*4 waves*iters*16*2048) and a 16-byte-per-lane stream copy. */
int gpfit_probe_mfma_f64(void* stream, double* scratch, int blocks, int iters);
int gpfit_probe_stream_copy(void* stream, const double* in, double* out, int64_t n_doubles);

#ifdef __cplusplus
}
#endif
#endif /* GPFIT_MI355X_H */
