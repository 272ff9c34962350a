// Host-side orchestration of the GP fit path on one MI355X: workspace context, the recursive
// blocked Cholesky built from MFMA GEMMs, and the fused unit of work (one evaluation of the
// reference's M-step closure, utils.py:2017-2112) in the original-basis Cholesky formulation
// (DESIGN.md section 3).
#include "context.h"
#include "gpfit_mi355x.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace gpfit {

// ------------------------------------------------------------------ per-launch profiling
static thread_local gpfit_ctx* g_prof = nullptr;
static thread_local void* g_main_sk_ws = nullptr;  // stream-K workspace of the context being evaluated

// tile-walk choices of the large launches (tuning knob GPFIT_WALKS = trsm,tmp,merge,T,Q,Rbase,H)
static const int* walks() {
  static int w[7] = {3, 2, 1, 9, 9, 2, 2};  // bit 3 = XCD-aware macro-tile schedule where the launch is large enough
                                          // (gemm_sched.hip), else the walk in the low bits
  static bool init = false;
  if (!init) {
    init = true;
    if (const char* e = getenv("GPFIT_WALKS")) {
      int v[7];
      if (sscanf(e, "%d,%d,%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6]) == 7)
        for (int i = 0; i < 7; ++i) w[i] = v[i];
    }
  }
  return w;
}

static hipEvent_t prof_event(gpfit_ctx* c) {
  hipEvent_t e;
  if (!c->ev_pool.empty()) {
    e = c->ev_pool.back();
    c->ev_pool.pop_back();
    return e;
  }
  (void)hipEventCreate(&e);
  return e;
}

void prof_begin(gpfit_ctx* c) {
  if (c->profile == 1) {
    g_prof = c;
    c->prof.clear();
  }
}

void prof_end(gpfit_ctx* c) {
  if (g_prof != c) return;
  g_prof = nullptr;
  (void)hipDeviceSynchronize();
  double ms[4] = {0, 0, 0, 0}, fl[4] = {0, 0, 0, 0}, cnt[4] = {0, 0, 0, 0};
  double big_ms = 0, big_fl = 0;
  for (auto& r : c->prof) {
    float t = 0.f;
    (void)hipEventElapsedTime(&t, r.a, r.b);
    if (r.kind == 0 && r.flops > big_fl) { big_fl = r.flops; big_ms = t; }
    ms[r.kind] += t;
    fl[r.kind] += r.flops;
    cnt[r.kind] += 1;
    c->ev_pool.push_back(r.a);
    c->ev_pool.push_back(r.b);
  }
  c->prof.clear();
  c->prof_out[0] = ms[0]; c->prof_out[1] = fl[0]; c->prof_out[2] = cnt[0];
  c->prof_out[3] = ms[1]; c->prof_out[4] = cnt[1];
  c->prof_out[5] = ms[2]; c->prof_out[6] = fl[2]; c->prof_out[7] = cnt[2];
  c->prof_out[8] = ms[3]; c->prof_out[9] = fl[3]; c->prof_out[10] = cnt[3];
  c->prof_out[11] = big_ms; c->prof_out[12] = big_fl;
}

ProfScope::ProfScope(hipStream_t s_, double flops_, int kind_) : s(s_), flops(flops_), kind(kind_) {
  if (g_prof) {
    a = prof_event(g_prof);
    (void)hipEventRecord(a, s);
  }
}
ProfScope::~ProfScope() {
  if (g_prof && a) {
    hipEvent_t b = prof_event(g_prof);
    (void)hipEventRecord(b, s);
    g_prof->prof.push_back({a, b, flops, kind});
  }
}

// flops actually executed by one GEMM launch (whole 128-tiles over each tile's k range)
template <typename R>
double gemm_flops(const GemmArgsT<R>& g) {
  const int T = gemm_pick_tile(g);
  const int tm = (g.M + T - 1) / T, tn = (g.N + T - 1) / T, r = TILE / T;
  double steps = 0;
  for (int ti = 0; ti < tm; ++ti)
    for (int tj = 0; tj < tn; ++tj) {
      if (g.out_lower && tj / r > ti / r) continue;
      int kb = 0, ke = g.K;
      if (g.a_tri == 1) ke = std::min(ke, ti * T + T);
      if (g.a_tri == 2) kb = std::max(kb, ti * T);
      if (g.b_tri == 1) kb = std::max(kb, tj * T);
      if (g.b_tri == 2) ke = std::min(ke, tj * T + T);
      if (ke > kb) steps += (ke - kb);
    }
  return 2.0 * T * T * steps * (g.nptr > 0 ? g.nptr : (g.batch > 0 ? g.batch : 1));
}

// ------------------------------------------------------------------ GEMM convenience
template <typename R>
static GemmArgsT<R> gemm_args(int a_kmajor, int b_kmajor, int M, int N, int K, double alpha, const R* A, int64_t lda,
                              const R* B, int64_t ldb, double beta, R* C, int64_t ldc, int out_lower, int a_tri,
                              int b_tri, int reverse = 0, int ws = 0, void* sk_ws = nullptr, int half_occ = 0) {
  GemmArgsT<R> g{};
  g.A = A; g.B = B; g.C = C;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K;
  g.alpha = alpha; g.beta = beta;
  g.a_kmajor = a_kmajor; g.b_kmajor = b_kmajor;
  g.out_lower = out_lower; g.a_tri = a_tri; g.b_tri = b_tri;
  g.batch = 1; g.split_k = 1; g.reverse = reverse; g.workspace = ws; g.sk_ws = sk_ws ? sk_ws : g_main_sk_ws;
  g.half_occ = half_occ;
  return g;
}

// tuning aid: GPFIT_GEMM_LOG=k lists every GEMM launch of the k-th evaluation of the process on stderr (shape,
// structure flags, problems in the batch, block tile, executed flops), in launch order -- to be paired with a
// kernel trace of the same run (scripts/trace_gemm_rates.py)
static int g_eval_count = 0;
static int gemm_log_eval() {
  static const int v = getenv("GPFIT_GEMM_LOG") ? atoi(getenv("GPFIT_GEMM_LOG")) : -1;
  return v;
}

template <typename R>
static int run_gemm(hipStream_t s, const GemmArgsT<R>& g, bool plain = false) {
  if (gemm_log_eval() >= 0 && g_eval_count == gemm_log_eval())
    fprintf(stderr, "[gpfit gemm] M %d N %d K %d atri %d btri %d lower %d nb %d tile %d ak %d bk %d epi %d flops %.6e\n", g.M, g.N, g.K,
            g.a_tri, g.b_tri, g.out_lower, g.nptr > 0 ? g.nptr : 1, gemm_pick_tile(g), g.a_kmajor, g.b_kmajor, g.epi, gemm_flops(g));
  // profile kind 0: the 128-tile kernel family (the dominant kernel), 3: the small-tile instances
  ProfScope ps(s, g_prof ? gemm_flops(g) : 0.0, (g_prof && gemm_pick_tile(g) != TILE) ? 3 : 0);
  return (plain || g.half_occ) ? launch_gemm_plain(g, s) : launch_gemm(g, s);  // plain: data-parallel, never stream-K
}

template <typename R>
static int gemm(hipStream_t s, int a_kmajor, int b_kmajor, int M, int N, int K, double alpha, const R* A,
                int64_t lda, const R* B, int64_t ldb, double beta, R* C, int64_t ldc, int out_lower, int a_tri,
                int b_tri, int reverse = 0, int ws = 0, void* sk_ws = nullptr, bool plain = false, int half_occ = 0) {
  return run_gemm(s, gemm_args<R>(a_kmajor, b_kmajor, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, out_lower, a_tri, b_tri,
                                  reverse, ws, sk_ws, half_occ), plain);
}

#define GP_TRY(expr)            \
  do {                          \
    int _rc = (expr);           \
    if (_rc != 0) return _rc;   \
  } while (0)

// C = alpha op(A) op(B) with the k range cut into `splits` slabs: the shapes of the truncated-rank closures whose
// output has few tiles but a long k (K_b = K~ B: 8192 x 512 x 8192; B^T X: 512 x 512 x 8192) fill the chip with
// 128-tiles only this way.  The slabs go to `partial` (splits x M x ldc elements) and are added in slab order by
// one pass (deterministic; C must be the contiguous block [M][ldc]).  splits <= 1: the plain launch.
template <typename R>
static int gemm_splitk(hipStream_t s, int a_kmajor, int b_kmajor, int M, int N, int K, double alpha, const R* A, int64_t lda,
                       const R* B, int64_t ldb, R* C, int64_t ldc, int splits, R* partial, int64_t partial_elems) {
  static const bool off = getenv("GPFIT_NO_SPLITK") != nullptr;   // tuning knob
  splits = (int)std::min<int64_t>(splits, partial_elems / std::max<int64_t>(1, (int64_t)M * ldc));   // what the scratch holds
  if (splits <= 1 || off || partial == nullptr)
    return gemm<R>(s, a_kmajor, b_kmajor, M, N, K, alpha, A, lda, B, ldb, 0.0, C, ldc, 0, 0, 0);
  GemmArgsT<R> g = gemm_args<R>(a_kmajor, b_kmajor, M, N, K, alpha, A, lda, B, ldb, 0.0, partial, ldc, 0, 0, 0);
  g.split_k = splits;
  g.sC = (int64_t)M * ldc;
  g.tile = TILE;
  GP_TRY(run_gemm(s, g));
  return launch_reduce_slices(partial, (int64_t)M * ldc, splits, C, (int64_t)M * ldc, s);
}
// slabs so that a product with few output tiles still launches about one and a half rounds of 128-tile workgroups,
// each with a k range of at least 256
static int splitk_for(int M, int N, int K) {
  const long tiles = (long)((M + TILE - 1) / TILE) * ((N + TILE - 1) / TILE);
  if (tiles >= 384) return 1;
  long sp = (768 + tiles - 1) / tiles;
  sp = std::min<long>(sp, std::max(1, K / 256));
  return (int)std::max<long>(1, sp);
}

// tuning knob (bit mask, default all): fused GEMM epilogues -- 1 Q's symmetrisation, 2 T's norm, 4 the H / Z21 update
// of the two-sided product
static int fused_epilogues() {
  static const int v = getenv("GPFIT_FUSED_EPI") ? atoi(getenv("GPFIT_FUSED_EPI")) : 7;
  return v;
}


// ------------------------------------------------------------------ recursive Cholesky (+ inverse)
template <typename R>
int potrf_rec(const CholBufsT<R>& B, int r0, int n, int need_inv, hipStream_t s) {
  const int64_t ld = B.ld;
  auto at = [&](R* base, int r, int c) { return base + (int64_t)r * ld + c; };
  if (n == TILE) {
    ProfScope ps(s, 0.0, 1);
    static const bool lds_leaf = getenv("GPFIT_LEAF_LDS") != nullptr;  // tuning knob: the LDS-resident leaf
    if (lds_leaf) return launch_chol_leaf(at(B.A, r0, r0), ld, at(B.L, r0, r0), ld, at(B.Li, r0, r0), ld, B.info, r0, s);
    return launch_chol_leaf_reg(at(B.A, r0, r0), ld, at(B.L, r0, r0), ld, at(B.Li, r0, r0), ld, B.info, r0, s);
  }
  const int k = n / TILE;
  const int n1 = ((k + 1) / 2) * TILE, n2 = n - n1;
  const int r1 = r0 + n1;
  GP_TRY(potrf_rec<R>(B, r0, n1, 1, s));
  if (B.mark_ev && r0 == 0 && n1 == B.mark_n) GP_HIP(hipEventRecord(B.mark_ev, s));
  // L21 = A21 * L11^-T       (trsm as a GEMM against the explicit inverse; op(B) = Li11^T is upper)
  GP_TRY(gemm<R>(s, 0, 0, n2, n1, n1, 1.0, at(B.A, r1, r0), ld, at(B.Li, r0, r0), ld, 0.0, at(B.L, r1, r0), ld, 0, 0, 2, walks()[0], B.ws, B.sk_ws, false, B.half_occ & 1));
  // Look-ahead: tmp = L21 * Li11, the first product of the inverse merge, needs nothing from the second
  // half, so it runs on the chain's side stream while that half is being factored (its leaves are
  // latency-bound and leave the chip to it).  Possible because the leaf shares a CU with GEMM workgroups.
  hipEvent_t joined = nullptr;
  if (need_inv == 1 && B.ctx && B.side_min > 0 && n >= B.side_min && B.ctx->side[B.chain]) {
    gpfit_ctx* c = B.ctx;
    auto next_event = [&]() {
      auto& pool = c->side_ev[B.chain];
      int& nx = c->side_ev_next[B.chain];
      if (nx == (int)pool.size()) {
        hipEvent_t e = nullptr;
        (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        pool.push_back(e);
      }
      return pool[nx++];
    };
    hipStream_t side = c->side[B.chain];
    hipEvent_t fork = next_event();
    joined = next_event();
    GP_HIP(hipEventRecord(fork, s));
    GP_HIP(hipStreamWaitEvent(side, fork, 0));
    GP_TRY(gemm<R>(side, 0, 1, n2, n1, n1, 1.0, at(B.L, r1, r0), ld, at(B.Li, r0, r0), ld, 0.0, at(B.Tmp, r1, r0), ld, 0, 0, 1, walks()[1], 2 + B.chain, c->sk_ws[2 + B.chain], false, (B.half_occ >> 1) & 1));
    GP_HIP(hipEventRecord(joined, side));
  }
  // A22 -= L21 L21^T          (syrk, lower tiles only); on the latency-bound levels tmp = L21 * Li11 shares its launch
  // (potrf_lockstep below does the same: the two routes stay launch-for-launch the same products)
  bool merged = false;
  if (need_inv == 1 && !joined && !(B.half_occ & 1)) {
    static const bool no_batch = getenv("GPFIT_NO_BATCH") != nullptr;
    auto one = [&](GemmArgsT<R> g) { g.nptr = 1; g.batch = 1; g.Ap[0] = g.A; g.Bp[0] = g.B; g.Cp[0] = g.C; return g; };
    const GemmArgsT<R> g2 = one(gemm_args<R>(0, 0, n2, n2, n1, -1.0, at(B.L, r1, r0), ld, at(B.L, r1, r0), ld, 1.0, at(B.A, r1, r1), ld,
                                             1, 0, 0, 0, B.ws, B.sk_ws));
    const GemmArgsT<R> g3 = one(gemm_args<R>(0, 1, n2, n1, n1, 1.0, at(B.L, r1, r0), ld, at(B.Li, r0, r0), ld, 0.0, at(B.Tmp, r1, r0), ld,
                                             0, 0, 1, walks()[1], B.ws, B.sk_ws));
    if (!no_batch && gemm_pair_ok(g2, g3)) {
      ProfScope ps(s, g_prof ? gemm_flops(g2) + gemm_flops(g3) : 0.0, 3);
      GP_TRY(launch_gemm_pair(g2, g3, s));
      merged = true;
    }
  }
  if (!merged)
    GP_TRY(gemm<R>(s, 0, 0, n2, n2, n1, -1.0, at(B.L, r1, r0), ld, at(B.L, r1, r0), ld, 1.0, at(B.A, r1, r1), ld, 1, 0, 0, 0, B.ws, B.sk_ws, false, B.half_occ & 1));
  GP_TRY(potrf_rec<R>(B, r1, n2, need_inv ? 1 : 0, s));
  if (need_inv == 1) {
    // Li21 = -Li22 * (L21 * Li11)
    if (joined) GP_HIP(hipStreamWaitEvent(s, joined, 0));
    else if (!merged) GP_TRY(gemm<R>(s, 0, 1, n2, n1, n1, 1.0, at(B.L, r1, r0), ld, at(B.Li, r0, r0), ld, 0.0, at(B.Tmp, r1, r0), ld, 0, 0, 1, walks()[1], B.ws, B.sk_ws, false, B.half_occ & 1));
    GP_TRY(gemm<R>(s, 0, 1, n2, n1, n2, -1.0, at(B.Li, r1, r1), ld, at(B.Tmp, r1, r0), ld, 0.0, at(B.Li, r1, r0), ld, 0, 1, 0, walks()[2], B.ws, B.sk_ws, false, B.half_occ & 1));
  }
  return 0;
}

template int potrf_rec<double>(const CholBufsT<double>&, int, int, int, hipStream_t);
template int potrf_rec<float>(const CholBufsT<float>&, int, int, int, hipStream_t);

// ------------------------------------------------------------------ lock-step recursion over several chains
// The same recursion for nb matrices of the same size at once (CholBatchT, context.h): the K~ and V chains of one
// unit, or the chains of several independent units.  Every level whose launches cannot fill the chip -- the
// leaves and the products of the small blocks, i.e. the latency-bound bottom of the recursion -- is ONE launch
// for all chains (a pointer batch, GemmArgsT::nptr / LeafBatchT): the number of kernel boundaries on the critical
// path no longer grows with the number of chains and every small launch has nb times the workgroups.  Products
// that are 128-tile launches for a single chain are issued chain by chain through the ordinary launcher, with
// its stream-K / XCD-aware schedules.  Results are bit-identical to potrf_rec: a chain's launches are the same
// products in the same order; the batched ones are data-parallel launches exactly where potrf_rec's are (stream-K
// only ever applies to 128-tile launches of a single problem, which take the same path here), and every
// data-parallel instance sums k in ascending order per element whatever block tile the launcher picks.
// need[b]: chain b needs the inverse of its block at this node (as need_inv of potrf_rec).
// cnt problems of one shape, problem i on (Ap[i], Bp[i], Cp[i]): one pointer-batched launch, unless the product
// is a 128-tile launch already for a single problem -- then problem by problem through the ordinary launcher.
template <typename R>
static int gemm_list(hipStream_t s, int cnt, const R* const* Ap, const R* const* Bp, R* const* Cp, int a_kmajor,
                     int b_kmajor, int M, int N, int K, double alpha, int64_t lda, int64_t ldb, double beta, int64_t ldc,
                     int out_lower, int a_tri, int b_tri, int reverse = 0, int ws_id = 0, void* sk_ws = nullptr,
                     int epi = 0, R* const* auxp = nullptr, double* const* sumsqp = nullptr, bool* epi_done = nullptr,
                     int* sumsq_entries = nullptr) {
  if (epi_done) *epi_done = false;
  if (sumsq_entries) *sumsq_entries = 0;
  if (cnt <= 0) return 0;
  if (cnt > GEMM_MAXB) {
    set_error("gemm_list: more problems than a pointer batch holds");
    return -3;
  }
  GemmArgsT<R> g = gemm_args<R>(a_kmajor, b_kmajor, M, N, K, alpha, Ap[0], lda, Bp[0], ldb, beta, Cp[0], ldc, out_lower,
                                a_tri, b_tri, reverse, ws_id, sk_ws);
  static const bool no_batch = getenv("GPFIT_NO_BATCH") != nullptr;   // tuning knob: every product on its own
  // epi: fused epilogue wanted (common.h); *epi_done says whether the launches carried it -- all of them or none
  // (the caller runs the separate passes otherwise)
  // Large-tile products normally go unit by unit through the ordinary launcher (balanced schedules).  Lists that
  // carry an epilogue (T and Q of a group) may share ONE data-parallel 128-tile launch instead once there are enough
  // of them to fill the chip for many rounds (GPFIT_LIST_T128 = smallest such count, 0 = never): no stream-K fix-up,
  // no partial tiles, and the epilogues (tile norms, mirrored store) ride along -- a single unit's stream-K launch
  // cannot carry them.
  static const int list_t128 = getenv("GPFIT_LIST_T128") ? atoi(getenv("GPFIT_LIST_T128")) : 0;
  static const int list_t128_dim = getenv("GPFIT_LIST_T128_DIM") ? atoi(getenv("GPFIT_LIST_T128_DIM")) : 4096;
  const bool share_t128 = epi != 0 && list_t128 > 0 && cnt >= list_t128 && M <= list_t128_dim && N <= list_t128_dim;
  if (cnt == 1 || (gemm_pick_tile(g) == TILE && !share_t128) || no_batch) {
    bool fused = epi != 0;
    for (int i = 0; i < cnt && fused; ++i) {
      g.A = Ap[i]; g.B = Bp[i]; g.C = Cp[i];
      g.epi = epi; g.aux = auxp ? auxp[i] : nullptr; g.sumsq = sumsqp ? sumsqp[i] : nullptr;
      fused = gemm_epilogue_ok(g);
    }
    for (int i = 0; i < cnt; ++i) {
      g.A = Ap[i]; g.B = Bp[i]; g.C = Cp[i];
      g.epi = fused ? epi : 0; g.aux = (fused && auxp) ? auxp[i] : nullptr; g.sumsq = (fused && sumsqp) ? sumsqp[i] : nullptr;
      GP_TRY(run_gemm(s, g));
    }
    if (epi_done) *epi_done = fused;
    if (fused && sumsq_entries) {
      g.A = Ap[0]; g.B = Bp[0]; g.C = Cp[0];
      g.epi = epi; g.aux = auxp ? auxp[0] : nullptr; g.sumsq = sumsqp ? sumsqp[0] : nullptr;
      *sumsq_entries = gemm_sumsq_entries(g);
    }
    return 0;
  }
  g.nptr = cnt;
  g.batch = cnt;
  for (int i = 0; i < cnt; ++i) {
    g.Ap[i] = Ap[i]; g.Bp[i] = Bp[i]; g.Cp[i] = Cp[i];
    g.auxp[i] = auxp ? auxp[i] : nullptr; g.sumsqp[i] = sumsqp ? sumsqp[i] : nullptr;
  }
  g.epi = epi;
  if (epi && !gemm_epilogue_ok(g)) g.epi = 0;
  if (epi_done) *epi_done = g.epi != 0;
  if (g.epi && sumsq_entries) *sumsq_entries = gemm_sumsq_entries(g);
  return run_gemm(s, g);
}

template <typename R>
static int bgemm(const CholBatchT<R>& B, hipStream_t s, uint32_t mask, int a_kmajor, int b_kmajor, int M, int N, int K,
                 double alpha, R* const* Ab, int64_t offA, R* const* Bb, int64_t offB, double beta, R* const* Cb,
                 int64_t offC, int out_lower, int a_tri, int b_tri, int reverse, int ws_id, void* sk_ws) {
  const R* Ap[GEMM_MAXB];
  const R* Bp[GEMM_MAXB];
  R* Cp[GEMM_MAXB];
  int cnt = 0;
  for (int b = 0; b < B.nb; ++b)
    if (mask & (1u << b)) {
      Ap[cnt] = Ab[b] + offA; Bp[cnt] = Bb[b] + offB; Cp[cnt] = Cb[b] + offC;
      ++cnt;
    }
  return gemm_list<R>(s, cnt, Ap, Bp, Cp, a_kmajor, b_kmajor, M, N, K, alpha, B.ld, B.ld, beta, B.ld, out_lower, a_tri,
                      b_tri, reverse, ws_id, sk_ws);
}

// the pointer batch of the chains in `mask` (as bgemm builds it), for launches that take two batches at once
template <typename R>
static GemmArgsT<R> batch_args(const CholBatchT<R>& B, uint32_t mask, int a_kmajor, int b_kmajor, int M, int N, int K, double alpha,
                               R* const* Ab, int64_t offA, R* const* Bb, int64_t offB, double beta, R* const* Cb, int64_t offC,
                               int out_lower, int a_tri, int b_tri, int reverse) {
  GemmArgsT<R> g = gemm_args<R>(a_kmajor, b_kmajor, M, N, K, alpha, (const R*)nullptr, B.ld, (const R*)nullptr, B.ld, beta,
                                (R*)nullptr, B.ld, out_lower, a_tri, b_tri, reverse, B.ws, B.sk_ws);
  int cnt = 0;
  for (int b = 0; b < B.nb; ++b)
    if (mask & (1u << b)) {
      g.Ap[cnt] = Ab[b] + offA; g.Bp[cnt] = Bb[b] + offB; g.Cp[cnt] = Cb[b] + offC;
      ++cnt;
    }
  g.nptr = cnt;
  g.batch = cnt;
  if (cnt > 0) { g.A = g.Ap[0]; g.B = g.Bp[0]; g.C = g.Cp[0]; }
  return g;
}

template <typename R>
int potrf_lockstep(const CholBatchT<R>& B, int r0, int n, uint32_t need, hipStream_t s) {
  const int64_t ld = B.ld;
  const uint32_t all = (B.nb >= 32) ? 0xffffffffu : ((1u << B.nb) - 1u);
  auto off = [&](int r, int c) { return (int64_t)r * ld + c; };
  if (n == TILE) {
    ProfScope ps(s, 0.0, 1);
    LeafBatchT<R> bt{};
    bt.n = B.nb;
    for (int b = 0; b < B.nb; ++b) {
      bt.A[b] = B.A[b] + off(r0, r0); bt.L[b] = B.L[b] + off(r0, r0); bt.Li[b] = B.Li[b] + off(r0, r0);
      bt.info[b] = B.info[b];
    }
    bt.lda = bt.ldl = bt.ldi = ld;
    bt.info_base = r0;
    return launch_chol_leaf_batch(bt, s);
  }
  const int k = n / TILE;
  const int n1 = ((k + 1) / 2) * TILE, n2 = n - n1;
  const int r1 = r0 + n1;
  GP_TRY(potrf_lockstep<R>(B, r0, n1, all, s));
  // L21 = A21 * L11^-T
  GP_TRY(bgemm<R>(B, s, all, 0, 0, n2, n1, n1, 1.0, B.A, off(r1, r0), B.Li, off(r0, r0), 0.0, B.L, off(r1, r0), 0, 0, 2,
                  walks()[0], B.ws, B.sk_ws));
  // look-ahead of the inverse merge's first product on the side stream (as potrf_rec)
  hipEvent_t joined = nullptr;
#ifdef GPFIT_DEV
  // timing experiment (wrong results by design): what the unit would cost if the first product of every inverse
  // merge of a block of at least this size were hidden completely
  static const int dev_skip_tmp = getenv("GPFIT_DEV_SKIP_TMP") ? atoi(getenv("GPFIT_DEV_SKIP_TMP")) : 0;
  const bool skip_tmp = dev_skip_tmp > 0 && n >= dev_skip_tmp;
#else
  const bool skip_tmp = false;
#endif
  if (need && !skip_tmp && B.ctx && B.side_min > 0 && n >= B.side_min && B.ctx->side[B.chain]) {
    gpfit_ctx* c = B.ctx;
    auto next_event = [&]() {
      auto& pool = c->side_ev[B.chain];
      int& nx = c->side_ev_next[B.chain];
      if (nx == (int)pool.size()) {
        hipEvent_t e = nullptr;
        (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        pool.push_back(e);
      }
      return pool[nx++];
    };
    hipStream_t side = c->side[B.chain];
    hipEvent_t fork = next_event();
    joined = next_event();
    GP_HIP(hipEventRecord(fork, s));
    GP_HIP(hipStreamWaitEvent(side, fork, 0));
    GP_TRY(bgemm<R>(B, side, need, 0, 1, n2, n1, n1, 1.0, B.L, off(r1, r0), B.Li, off(r0, r0), 0.0, B.Tmp, off(r1, r0), 0, 0, 1,
                    walks()[1], 2 + B.chain, c->sk_ws[2 + B.chain]));
    GP_HIP(hipEventRecord(joined, side));
  }
  // A22 -= L21 L21^T.  On the latency-bound levels the first product of the inverse merge, L21 L11^-1 (it needs
  // L21 and L11^-1 only), rides in the same launch: one launch boundary less per node of the recursion.
  bool merged = false;
  if (need && !joined && !skip_tmp) {
    static const bool no_batch = getenv("GPFIT_NO_BATCH") != nullptr;
    const GemmArgsT<R> g2 = batch_args<R>(B, all, 0, 0, n2, n2, n1, -1.0, B.L, off(r1, r0), B.L, off(r1, r0), 1.0, B.A, off(r1, r1),
                                          1, 0, 0, 0);
    const GemmArgsT<R> g3 = batch_args<R>(B, need, 0, 1, n2, n1, n1, 1.0, B.L, off(r1, r0), B.Li, off(r0, r0), 0.0, B.Tmp,
                                          off(r1, r0), 0, 0, 1, walks()[1]);
    if (!no_batch && gemm_pair_ok(g2, g3)) {
      if (gemm_log_eval() >= 0 && g_eval_count == gemm_log_eval())
        fprintf(stderr, "[gpfit gemm] pair: M %d N %d K %d lower 1 nb %d + M %d N %d K %d btri 1 nb %d tile %d flops %.6e\n", g2.M, g2.N,
                g2.K, g2.nptr, g3.M, g3.N, g3.K, g3.nptr, gemm_pick_tile(g3), gemm_flops(g2) + gemm_flops(g3));
      ProfScope ps(s, g_prof ? gemm_flops(g2) + gemm_flops(g3) : 0.0, 3);
      GP_TRY(launch_gemm_pair(g2, g3, s));
      merged = true;
    }
  }
  if (!merged)
    GP_TRY(bgemm<R>(B, s, all, 0, 0, n2, n2, n1, -1.0, B.L, off(r1, r0), B.L, off(r1, r0), 1.0, B.A, off(r1, r1), 1, 0, 0, 0,
                    B.ws, B.sk_ws));
  GP_TRY(potrf_lockstep<R>(B, r1, n2, need, s));
  if (need) {
    if (joined) GP_HIP(hipStreamWaitEvent(s, joined, 0));
    else if (!merged && !skip_tmp)
      GP_TRY(bgemm<R>(B, s, need, 0, 1, n2, n1, n1, 1.0, B.L, off(r1, r0), B.Li, off(r0, r0), 0.0, B.Tmp, off(r1, r0), 0, 0,
                      1, walks()[1], B.ws, B.sk_ws));
    GP_TRY(bgemm<R>(B, s, need, 0, 1, n2, n1, n2, -1.0, B.Li, off(r1, r1), B.Tmp, off(r1, r0), 0.0, B.Li, off(r1, r0), 0, 1, 0,
                    walks()[2], B.ws, B.sk_ws));
  }
  return 0;
}

template int potrf_lockstep<double>(const CholBatchT<double>&, int, int, uint32_t, hipStream_t);
template int potrf_lockstep<float>(const CholBatchT<float>&, int, int, uint32_t, hipStream_t);

// ------------------------------------------------------------------ two-sided triangular product
// Wout (lower) = 1/2 Li^T Q Li on the n x n diagonal block at r0, Q symmetric (stored in full),
// Li lower triangular.  The direct route R = Q Li, W = Li^T R costs 4/3 n^3; splitting once,
//   Li = [A 0; B C],  H = Q21 A + 1/2 Q22 B,
//   W11 = A^T Q11 A + B^T H + H^T B,   W21 = C^T (H + 1/2 Q22 B),   W22 = C^T Q22 C,
// costs 3/4 n^3 plus the two half-size products (LAPACK's sygst idea), i.e. 13/12 n^3 with
// one level; blocks of n >= min_split (4096) are split again: 1.02 n^3 at n = 8192.  Z and H are n x n scratch matrices with the same leading dimension.
template <typename R>
struct TwoSidedBufs {
  const R* Q; const R* Li; R* W; R* Z; R* H; int64_t ld; int min_split;
  // optional: a side stream with two events; the top split then runs its two half-size diagonal products there,
  // beside the three large off-diagonal products on the main stream (they depend on nothing of their level)
  hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  void* side_sk_ws = nullptr;   // stream-K workspace of the launches issued on `side`
  void* sk_ws = nullptr;        // stream-K workspace of this block's own launches (nullptr: the main stream's)
};
// cnt diagonal blocks of size n (block i of problem b[i] at offset r0[i]) in lock step: the two half-size
// two-sided products a split leaves behind (W11's A^T Q11 A and W22) depend on nothing else of their level, so
// they -- and the sub-blocks of several units -- share their launches (gemm_list above): two 2048-sized problems
// are one 128-tile launch of 512 workgroups instead of two 64-tile launches.  Same products, same order per
// block: bit-identical to the block-by-block recursion.
template <typename R>
static int two_sided_list(int cnt, const TwoSidedBufs<R>* b, const int* r0, int n, hipStream_t s) {
  if (cnt <= 0) return 0;
  const int64_t ld = b[0].ld;
  auto at = [&](const R* base, int r, int c) { return const_cast<R*>(base) + (int64_t)r * ld + c; };
  const R* Ap[GEMM_MAXB];
  const R* Bp[GEMM_MAXB];
  R* Cp[GEMM_MAXB];
  const int k = n / TILE;
  if (n < b[0].min_split || k < 2) {
    static const int wbase_walk = getenv("GPFIT_WBASE_WALK") ? atoi(getenv("GPFIT_WBASE_WALK")) : 0;
    for (int i = 0; i < cnt; ++i) { Ap[i] = at(b[i].Q, r0[i], r0[i]); Bp[i] = at(b[i].Li, r0[i], r0[i]); Cp[i] = at(b[i].Z, r0[i], r0[i]); }
    GP_TRY(gemm_list<R>(s, cnt, Ap, Bp, Cp, 1, 1, n, n, n, 1.0, ld, ld, 0.0, ld, 0, 0, 1, walks()[5], 0, b[0].sk_ws));
    for (int i = 0; i < cnt; ++i) { Ap[i] = at(b[i].Li, r0[i], r0[i]); Bp[i] = at(b[i].Z, r0[i], r0[i]); Cp[i] = at(b[i].W, r0[i], r0[i]); }
    GP_TRY(gemm_list<R>(s, cnt, Ap, Bp, Cp, 1, 1, n, n, n, 0.5, ld, ld, 0.0, ld, 1, 2, 0, wbase_walk, 0, b[0].sk_ws));
    return 0;
  }
  const int n1 = ((k + 1) / 2) * TILE, n2 = n - n1;
  auto diagonal_blocks = [&](hipStream_t st) -> int {
    // 1/2 A^T Q11 A (into W11) and W22 = 1/2 C^T Q22 C, all of them in lock step when the halves are equal
    TwoSidedBufs<R> bb[GEMM_MAXB];
    int rr[GEMM_MAXB];
    if (n1 == n2 && 2 * cnt <= GEMM_MAXB) {
      for (int i = 0; i < cnt; ++i) {
        bb[2 * i] = b[i]; rr[2 * i] = r0[i];
        bb[2 * i + 1] = b[i]; rr[2 * i + 1] = r0[i] + n1;
        bb[2 * i].side = bb[2 * i + 1].side = nullptr;
        if (st != s) bb[2 * i].sk_ws = bb[2 * i + 1].sk_ws = b[i].side_sk_ws;
      }
      return two_sided_list<R>(2 * cnt, bb, rr, n1, st);
    }
    for (int i = 0; i < cnt; ++i) { bb[i] = b[i]; bb[i].side = nullptr; rr[i] = r0[i] + n1; if (st != s) bb[i].sk_ws = b[i].side_sk_ws; }
    GP_TRY(two_sided_list<R>(cnt, bb, r0, n1, st));
    return two_sided_list<R>(cnt, bb, rr, n2, st);
  };
  const bool forked = cnt == 1 && b[0].side != nullptr;
  if (forked) {
    GP_HIP(hipEventRecord(b[0].ev_fork, s));
    GP_HIP(hipStreamWaitEvent(b[0].side, b[0].ev_fork, 0));
    GP_TRY(diagonal_blocks(b[0].side));
    GP_HIP(hipEventRecord(b[0].ev_join, b[0].side));
  }
  // Z21 = 1/2 Q22 B
  for (int i = 0; i < cnt; ++i) {
    const int r1 = r0[i] + n1;
    Ap[i] = at(b[i].Q, r1, r1); Bp[i] = at(b[i].Li, r1, r0[i]); Cp[i] = at(b[i].Z, r1, r0[i]);
  }
  GP_TRY(gemm_list<R>(s, cnt, Ap, Bp, Cp, 1, 1, n2, n1, n2, 0.5, ld, ld, 0.0, ld, 0, 0, 0, 0, 0, b[0].sk_ws));
  // H = Q21 A + Z21 ;  Z21 = H + 1/2 Q22 B  (one launch with the dual-update epilogue: H = acc + Z21, Z21 += H;
  // where the launch cannot carry it, a copy, the product with beta = 1 and an axpby pass -- the same arithmetic)
  {
    R* Zp[GEMM_MAXB];
    for (int i = 0; i < cnt; ++i) {
      const int r1 = r0[i] + n1;
      Ap[i] = at(b[i].Q, r1, r0[i]); Bp[i] = at(b[i].Li, r0[i], r0[i]); Cp[i] = at(b[i].H, r1, r0[i]); Zp[i] = at(b[i].Z, r1, r0[i]);
    }
    // would the fused launch be possible?  (asked first: the unfused route must copy Z21 into H beforehand)
    bool fused = false;
    if (fused_epilogues() & 4) {
      GemmArgsT<R> g = gemm_args<R>(0, 1, n2, n1, n1, 1.0, Ap[0], ld, Bp[0], ld, 0.0, Cp[0], ld, 0, 0, 1, walks()[6], 0, b[0].sk_ws);
      g.epi = 4; g.aux = Zp[0];
      static const bool no_batch = getenv("GPFIT_NO_BATCH") != nullptr;
      if (cnt > 1 && gemm_pick_tile(g) != TILE && !no_batch) { g.nptr = cnt; g.batch = cnt; }
      fused = gemm_epilogue_ok(g);
    }
    if (fused) {
      bool done = false;
      GP_TRY(gemm_list<R>(s, cnt, Ap, Bp, Cp, 0, 1, n2, n1, n1, 1.0, ld, ld, 0.0, ld, 0, 0, 1, walks()[6], 0, b[0].sk_ws, 4, Zp, nullptr,
                          &done));
      if (!done) {
        set_error("two_sided: the dual-update epilogue was announced but not carried");
        return -100;
      }
    } else {
      for (int i = 0; i < cnt; ++i)
        GP_HIP(hipMemcpy2DAsync(Cp[i], (size_t)ld * sizeof(R), Zp[i], (size_t)ld * sizeof(R), (size_t)n1 * sizeof(R), (size_t)n2,
                                hipMemcpyDeviceToDevice, s));
      GP_TRY(gemm_list<R>(s, cnt, Ap, Bp, Cp, 0, 1, n2, n1, n1, 1.0, ld, ld, 1.0, ld, 0, 0, 1, walks()[6], 0, b[0].sk_ws));
      for (int i = 0; i < cnt; ++i) GP_TRY(launch_axpby_block<R>(Zp[i], ld, Cp[i], ld, n2, n1, 1.0, 1.0, s));
    }
  }
  // W21 = 1/2 C^T Z21
  static const int w21_walk = getenv("GPFIT_W21_WALK") ? atoi(getenv("GPFIT_W21_WALK")) : 0;
  for (int i = 0; i < cnt; ++i) {
    const int r1 = r0[i] + n1;
    Ap[i] = at(b[i].Li, r1, r1); Bp[i] = at(b[i].Z, r1, r0[i]); Cp[i] = at(b[i].W, r1, r0[i]);
  }
  GP_TRY(gemm_list<R>(s, cnt, Ap, Bp, Cp, 1, 1, n2, n1, n2, 0.5, ld, ld, 0.0, ld, 0, 2, 0, w21_walk, 0, b[0].sk_ws));
  if (forked) GP_HIP(hipStreamWaitEvent(s, b[0].ev_join, 0));
  else GP_TRY(diagonal_blocks(s));
  // W11 += 1/2 (B^T H + H^T B)   (lower tiles)
  for (int i = 0; i < cnt; ++i) {
    const int r1 = r0[i] + n1;
    Ap[i] = at(b[i].Li, r1, r0[i]); Bp[i] = at(b[i].H, r1, r0[i]); Cp[i] = at(b[i].W, r0[i], r0[i]);
  }
  GP_TRY(gemm_list<R>(s, cnt, Ap, Bp, Cp, 1, 1, n1, n1, n2, 0.5, ld, ld, 1.0, ld, 1, 0, 0, 0, 0, b[0].sk_ws));
  for (int i = 0; i < cnt; ++i) {
    const int r1 = r0[i] + n1;
    Ap[i] = at(b[i].H, r1, r0[i]); Bp[i] = at(b[i].Li, r1, r0[i]);
  }
  GP_TRY(gemm_list<R>(s, cnt, Ap, Bp, Cp, 1, 1, n1, n1, n2, 0.5, ld, ld, 1.0, ld, 1, 0, 0, 0, 0, b[0].sk_ws));
  return 0;
}
template <typename R>
static int two_sided(const TwoSidedBufs<R>& b, int r0, int n, hipStream_t s) {
  return two_sided_list<R>(1, &b, &r0, n, s);
}

// ------------------------------------------------------------------ host pieces of localker
static double lin_pm1_host(int i, int n) {
  if (n <= 1) return -1.0;
  const double step = 2.0 / (double)(n - 1);
  return (i < n / 2) ? std::fma(step, (double)i, -1.0) : std::fma(-step, (double)(n - 1 - i), 1.0);
}

static int compute_mask(const double* theta, int n_rows, int n_cols, uint8_t* mask, int* pix) {
  const double eb = std::exp(theta[3]);
  int d = 0;
  for (int p = 0; p < n_rows * n_cols; ++p) {
    const double x = lin_pm1_host(p % n_cols, n_cols), y = lin_pm1_host(p / n_cols, n_rows);
    const double dx = x - theta[1], dy = y - theta[2];
    const double alpha = std::exp(-eb * (dx * dx + dy * dy));  // utils.py:880-881
    const bool keep = alpha >= 0.001;                          // utils.py:883
    if (mask) mask[p] = keep ? 1 : 0;
    if (keep) {
      if (pix) pix[d] = p;
      ++d;
    }
  }
  return d;
}

static Theta make_theta(const double* t) {
  Theta th;
  th.sigma0 = t[0]; th.eps0x = t[1]; th.eps0y = t[2]; th.logbeta = t[3]; th.logrho = t[4]; th.amp = t[5];
  th.eb = std::exp(t[3]);
  th.er = std::exp(t[4]);
  return th;
}

static int check_limits(const double* theta, const double* lower, const double* upper) {
  static const char* names[6] = {"sigma_0", "eps_0x", "eps_0y", "-2log2beta", "-log2rho2", "Amp"};
  for (int i = 0; i < 6; ++i) {
    if (!(lower[i] <= theta[i] && theta[i] <= upper[i])) {  // utils.py:866, 2023 (NaN fails too)
      char buf[256];
      snprintf(buf, sizeof buf, "%s = %.4f is not within the limits of %g and %g", names[i], theta[i], lower[i],
               upper[i]);
      set_error(buf);
      return -2;
    }
  }
  return 0;
}

// Side streams of the two chains (look-ahead products, the diagonal blocks of the two-sided product): off the
// critical path, lowest priority.  Created when a synchronous evaluation first wants them -- contexts that only
// ever serve grouped / asynchronous evaluations never do, and every stream a process creates is one more
// claimant of the few hardware queues.
static int ensure_side_streams(gpfit_ctx* c) {
  if (c->side[0]) return 0;
  int least = 0, greatest = 0;
  GP_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
  for (int i = 0; i < 2; ++i) GP_HIP(hipStreamCreateWithPriority(&c->side[i], hipStreamNonBlocking, least));
  return 0;
}

template <typename T>
static int dev_alloc(gpfit_ctx* c, T** p, size_t count) {
  void* q = nullptr;
  GP_HIP(hipMalloc(&q, count * sizeof(T)));
  c->allocs.push_back(q);
  *p = (T*)q;
  return 0;
}

// Everything after the join of the two factorisation chains: T = L^-1 L_V and its norm, and (with
// gradients) Q = I - T T^T, the two-sided product W, the adjoint pass and the pull-back to the metric.
// Templated separately from the first half of the unit so that the mixed-precision mode (fp64
// factorisations, fp32 gradient products) can run it on single-precision copies.
template <typename R>
struct PostJoin {
  const R *Li, *LV, *Cos, *bv, *q, *wl, *Xm, *Cmat;           // inputs
  R *T, *W, *Z, *H, *A, *Y, *tvec, *Mpart, *Mmat;             // work matrices (np^2), Y [np][dp], Mpart / Mmat
};
// cnt units at once (the units of a group, gpfit_fit_eval_batch; cnt = 1: the single unit): every product goes
// through gemm_list / two_sided_list -- one pointer-batched launch where a single unit's product cannot fill the
// chip, unit by unit through the ordinary launcher (balanced schedules, fused epilogues) where it can -- and the
// element-wise passes run unit by unit, all on ONE stream: a stream that waits on an event is not free on this
// runtime (every queue with a pending barrier packet slows the dispatch of the others), so a group gets its
// concurrency from batched launches, not from streams.
template <typename R, typename PhaseFn>
static int post_join_list(int cnt, gpfit_ctx* const* cs, const PostJoin<R>* a, const Theta* th, int n, int np, const int* d,
                          const int* dp, int n_rows, int n_cols, int want_grad, hipStream_t s, PhaseFn&& phase, bool side_ok) {
  const int64_t ld = np;
  const R* Ap[GEMM_MAXB];
  const R* Bp[GEMM_MAXB];
  R* Cp[GEMM_MAXB];
  double* Sp[GEMM_MAXB];
  if (cnt <= 0 || cnt > GEMM_MAXB) return cnt == 0 ? 0 : -3;
  // T = L^-1 L_V (lower x lower -> lower);  tr(K~^-1 V) = ||T||_F^2
  {
    for (int i = 0; i < cnt; ++i) { Ap[i] = a[i].Li; Bp[i] = a[i].LV; Cp[i] = a[i].T; Sp[i] = cs[i]->frob_part; }
    bool normed = false;
    int norm_entries = 0;
    // the tiles leave their sums of squares behind (no separate pass over T) where the launch can carry the epilogue
    GP_TRY(gemm_list<R>(s, cnt, Ap, Bp, Cp, 0, 1, np, np, np, 1.0, ld, ld, 0.0, ld, 1, 1, 1, walks()[3], 0, cs[0]->sk_ws[0],
                        (fused_epilogues() & 2) ? 2 : 0, nullptr, Sp, &normed, &norm_entries));
    for (int i = 0; i < cnt; ++i) {
      if (normed) GP_TRY(launch_frob_finish(cs[i]->frob_part, norm_entries, cs[i]->scal + 5, s));
      else GP_TRY(launch_frob_lower(a[i].T, ld, np, cs[i]->scal + 5, cs[i]->frob_part, s));
    }
  }
  phase(4, s);
  if (!want_grad) return 0;
  // W = 1/2 (K~^-1 - K~^-1 V K~^-1) = 1/2 Li^T (I - T T^T) Li        (T = L^-1 L_V)
  //   Q = I - T T^T   lower x upper, lower tiles only          N^3/3
  //   W = 1/2 Li^T Q Li  two-sided product (two_sided_list)    13/12 N^3 with one split
  //                      (direct: R = Q Li, W = 1/2 Li^T R     4/3 N^3)
  {
    // T T^T: every tile of a tile column has the same k range [0, col + 128).  XCD-aware macro-tile schedule for
    // a single large unit (2.97 ms at N = 8192 in the fit; the column-major heavy-first data-parallel walk 3.02,
    // stream-K 3.2)
    for (int i = 0; i < cnt; ++i) { Ap[i] = a[i].T; Bp[i] = a[i].T; Cp[i] = a[i].W; }
    bool mirrored = false;
    GP_TRY(gemm_list<R>(s, cnt, Ap, Bp, Cp, 0, 0, np, np, np, -1.0, ld, ld, 0.0, ld, 1, 1, 2, walks()[4], 0, cs[0]->sk_ws[0],
                        (fused_epilogues() & 1) ? 1 : 0, nullptr, nullptr, &mirrored));
    for (int i = 0; i < cnt; ++i) {
      GP_TRY(launch_add_diag(a[i].W, ld, np, 1.0, s));
      if (!mirrored) GP_TRY(launch_symmetrize(a[i].W, ld, np, s));   // otherwise the tiles stored their transposes
    }
  }
  phase(5, s);
  {
    static const int ts_min = getenv("GPFIT_TS_MIN") ? atoi(getenv("GPFIT_TS_MIN")) : 4096;
    TwoSidedBufs<R> tb[GEMM_MAXB];
    int r0[GEMM_MAXB];
    for (int i = 0; i < cnt; ++i) {
      tb[i] = TwoSidedBufs<R>{a[i].W, a[i].Li, a[i].T, a[i].Z, a[i].H, ld, ts_min > 0 ? ts_min : (1 << 30)};
      tb[i].sk_ws = cs[0]->sk_ws[0];
      r0[i] = 0;
    }
    // tuning knob: the half-size diagonal products of the top split on the chain-0 side stream (a single unit
    // evaluated synchronously only, as the look-ahead of the factorisation)
    static const int ts_side = getenv("GPFIT_TS_SIDE") ? atoi(getenv("GPFIT_TS_SIDE")) : 0;
    gpfit_ctx* c = cs[0];
    if (cnt == 1 && ts_side && side_ok && c->side[0]) {
      auto next_event = [&]() {
        auto& pool = c->side_ev[0];
        int& nx = c->side_ev_next[0];
        if (nx == (int)pool.size()) {
          hipEvent_t e = nullptr;
          (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
          pool.push_back(e);
        }
        return pool[nx++];
      };
      tb[0].side = c->side[0]; tb[0].ev_fork = next_event(); tb[0].ev_join = next_event(); tb[0].side_sk_ws = c->sk_ws[2];
    }
    GP_TRY(two_sided_list<R>(cnt, tb, r0, np, s));
  }
  phase(6, s);
  const int t64 = np / 64;
  for (int i = 0; i < cnt; ++i) {
    gpfit_ctx* c = cs[i];
    GP_TRY(launch_adjoint(a[i].T, a[i].Cos, ld, a[i].bv, a[i].q, n, np, a[i].A, c->upart, c->vpart, c->sumA_part, s));
    GP_TRY(launch_adjoint_reduce(c->upart, c->vpart, c->sumA_part, t64, t64 * (t64 + 1) / 2, a[i].q, a[i].wl, n, np,
                                 a[i].tvec, c->rpad, c->scal + 7, s));
  }
  // pull the contraction with dK~ back to the d x d metric: M = X^T (Aw + diag t) X   (units with the same
  // masked pixel count share the launch)
  {
    bool same_dp = true;
    for (int i = 1; i < cnt; ++i) same_dp = same_dp && dp[i] == dp[0];
    for (int i = 0; i < cnt; ++i) { Ap[i] = a[i].A; Bp[i] = a[i].Xm; Cp[i] = a[i].Y; }
    if (same_dp) GP_TRY(gemm_list<R>(s, cnt, Ap, Bp, Cp, 1, 1, np, dp[0], np, 1.0, ld, dp[0], 0.0, dp[0], 0, 0, 0, 0, 0, cs[0]->sk_ws[0]));
    else
      for (int i = 0; i < cnt; ++i) GP_TRY(gemm<R>(s, 1, 1, np, dp[i], np, 1.0, a[i].A, ld, a[i].Xm, dp[i], 0.0, a[i].Y, dp[i], 0, 0, 0));
  }
  for (int i = 0; i < cnt; ++i) {
    gpfit_ctx* c = cs[i];
    GP_TRY(launch_rowscale_add(a[i].Y, dp[i], a[i].Xm, dp[i], a[i].tvec, np, dp[i], s));
    GemmArgsT<R> g{};
    g.A = a[i].Xm; g.B = a[i].Y; g.C = a[i].Mpart;
    g.lda = dp[i]; g.ldb = dp[i]; g.ldc = dp[i];
    g.M = dp[i]; g.N = dp[i]; g.K = np;
    g.alpha = 1.0; g.beta = 0.0; g.a_kmajor = 1; g.b_kmajor = 1;
    g.batch = 1; g.split_k = c->split_k_M; g.sC = (int64_t)dp[i] * dp[i];
    {
      ProfScope ps(s, g_prof ? gemm_flops(g) : 0.0, (g_prof && gemm_pick_tile(g) != TILE) ? 3 : 0);
      GP_TRY(launch_gemm(g, s));
    }
    GP_TRY(launch_reduce_slices(a[i].Mpart, (int64_t)dp[i] * dp[i], c->split_k_M, a[i].Mmat, (int64_t)dp[i] * dp[i], s));
    GP_TRY(launch_metric_contract(th[i], c->pix, d[i], n_rows, n_cols, a[i].Cmat, dp[i], a[i].Mmat, dp[i], c->scal + 10, c->upart,
                                  c->info + 3, s));
  }
  return 0;
}
template <typename R, typename PhaseFn>
static int post_join(gpfit_ctx* c, const PostJoin<R>& a, const Theta& th, int n, int np, int d, int dp, int n_rows,
                     int n_cols, int want_grad, hipStream_t s, PhaseFn&& phase, bool side_ok) {
  return post_join_list<R>(1, &c, &a, &th, n, np, &d, &dp, n_rows, n_cols, want_grad, s, phase, side_ok);
}

// The fused unit of work, templated on the scalar type of the device data: fp64 is the
// reference's precision (headline); fp32 serves the hyperparameter-grid configuration
// (BASELINE configs[4]) -- every matrix, factorisation and GEMM in fp32 on v_mfma_f32_16x16x4_f32,
// scalars and reductions accumulated in fp64.
int fit_eval_finish(gpfit_ctx* c, double* out_host);

template <typename R>
static int fit_eval_impl(gpfit_ctx* c, void* stream, const double* theta, const double* lower, const double* upper,
                         int n_rows, int n_cols, const R* X, int64_t ldx, int64_t N, const R* r, const R* m,
                         const R* V, int64_t ldv, double logA, double lambda0, int want_grad, double* out_host,
                         R* lam_m_out, R* lam_var_out, R* f_out) {
  auto RP = [](double* b) { return reinterpret_cast<R*>(b); };  // workspace is allocated for fp64
  if (!c || !theta || !X || !r || !m || !V || !out_host || N <= 0) {
    set_error("gpfit_fit_eval: bad argument");
    return -3;
  }
  GP_CTX_ENTER(c, "gpfit_fit_eval");
  const double inf = std::numeric_limits<double>::infinity();
  if (lower && upper && check_limits(theta, lower, upper) != 0) {
    // utils.py:2020-2028: out-of-box theta -> infinite loss and infinite gradients
    out_host[0] = inf;
    out_host[1] = out_host[2] = std::numeric_limits<double>::quiet_NaN();
    for (int i = 0; i < 6; ++i) out_host[3 + i] = inf;
    return -2;
  }
  hipStream_t s = (hipStream_t)stream, sa = c->aux;
  if (getenv("GPFIT_SINGLE_STREAM")) sa = s;
  const int n = (int)N, np = (int)round_up(N, TILE);
  const int dfull = n_rows * n_cols;
  if (np > c->np_cap || dfull > c->dfull_cap) {
    set_error("gpfit_fit_eval: problem larger than the context capacity");
    return -3;
  }
  const int d = compute_mask(theta, n_rows, n_cols, nullptr, c->pix_host);
  const int dp = (int)round_up(d, 32);
  if (d <= 0 || dp > c->dp_cap) {
    set_error("gpfit_fit_eval: masked pixel count is zero or exceeds the context capacity");
    return -3;
  }
  const Theta th = make_theta(theta);
  const double s0sq = th.sigma0 * th.sigma0;
  const double A = std::exp(logA);
  const int64_t ld = np;
  c->cur_n = n; c->cur_np = np; c->cur_d = d; c->cur_dp = dp;

  ++g_eval_count;
  const auto t_host0 = std::chrono::steady_clock::now();
  auto phase = [&](int i, hipStream_t st) {
    if (c->profile != 2) return;
    if (!c->phase_ev[i]) (void)hipEventCreate(&c->phase_ev[i]);
    (void)hipEventRecord(c->phase_ev[i], st);
  };
  c->phase_valid = false;
  phase(0, s);
  g_main_sk_ws = c->sk_ws[0];
  prof_begin(c);
  struct ProfGuard { gpfit_ctx* c; ~ProfGuard() { prof_end(c); } } prof_guard{c};
  GP_HIP(hipMemsetAsync(c->info, 0, 4 * sizeof(int), s));
  GP_HIP(hipMemcpyAsync(c->pix, c->pix_host, (size_t)d * sizeof(int), hipMemcpyHostToDevice, s));
  GP_HIP(hipMemsetAsync(RP(c->mpad), 0, (size_t)np * sizeof(R), s));
  GP_HIP(hipMemcpyAsync(RP(c->mpad), m, (size_t)n * sizeof(R), hipMemcpyDeviceToDevice, s));

  // ---- aux stream: Cholesky of V (only log|V| and L_V are needed; no full inverse).
  // Opt-in reuse (flag bit 1 of want_grad): the caller promises V is the matrix of the previous
  // call on this context (V is constant during an M-step, utils.py:2016-2114), so L_V and
  // log|V| are kept.  bench.py never sets it: the unit of work includes this factorisation.
  const bool reuse_V = (want_grad & 2) && c->lv_valid && c->lv_n == n && c->lv_bytes == (int)sizeof(R);
  const bool async_call = (want_grad & 4) != 0;
  const bool mixed_grad = (want_grad & 8) != 0 && sizeof(R) == 8 && (want_grad & 1);
  want_grad &= 1;
  // The V chain starts together with potrf(K~), not at the top of the call: the two recursions have
  // the same shape, so started together their leaf phases and their large GEMMs coincide -- a large
  // GEMM of one chain otherwise keeps every CU occupied and the other chain's leaf (133 KiB of LDS)
  // waits for its tail (32.5 -> 31.8 ms/fit).  GPFIT_FORK_EARLY restores the old order (tuning knob).
  static const bool fork_late = getenv("GPFIT_FORK_EARLY") == nullptr;
  if (!fork_late) {
    GP_HIP(hipEventRecord(c->ev_fork, s));
    GP_HIP(hipStreamWaitEvent(sa, c->ev_fork, 0));
  }
  // tuning knob: smallest block whose inverse-merge product goes to the side stream (0 = never)
  // (only for synchronous calls: with several units in flight on several contexts the chip is busy
  // anyway and four more streams per context oversubscribe the hardware queues -- 64 cells x N = 4096,
  // four in flight: 161 -> 118 cells/s with side streams)
  static const int side_min_env = getenv("GPFIT_SIDE_MIN") ? atoi(getenv("GPFIT_SIDE_MIN")) : 1024;
  const int side_min = async_call ? 0 : side_min_env;
  if (!async_call) GP_TRY(ensure_side_streams(c));
  // tuning knob (bit mask): 1 = the V chain's own 128-tile launches at one workgroup per CU, 2 = the
  // side-stream products of both chains
  static const int half_occ = getenv("GPFIT_HALF_OCC") ? atoi(getenv("GPFIT_HALF_OCC")) : 0;
  c->side_ev_next[0] = c->side_ev_next[1] = 0;
  auto enqueue_v_chain = [&]() -> int {
  if (!reuse_V) {
      c->lv_valid = false; c->lv32_valid = false;
      GP_TRY(launch_pack_lower(V, ldv, n, RP(c->Vbuf), ld, np, sa));
      CholBufsT<R> bv{RP(c->Vbuf), RP(c->LVbuf), RP(c->LiVbuf), RP(c->TmpV), ld, c->info + 1, 1, c->sk_ws[1], c, 1, side_min, half_occ & 3};
      GP_TRY(potrf_rec<R>(bv, 0, np, false, sa));
      GP_TRY(launch_logdet(RP(c->LVbuf), ld, n, c->scal + 40, sa));
    }
    phase(3, sa);
    GP_HIP(hipEventRecord(c->ev_join, sa));
    return 0;
  };
  // Lock-step mode (default; GPFIT_LOCKSTEP=0 restores the two free-running chains): the V chain is not a
  // stream of its own -- both matrices go through ONE recursion (potrf_lockstep) whose latency-bound levels are
  // shared launches.  Only V's packing runs on the aux stream, beside the kernel build.
  static const int lockstep_env = getenv("GPFIT_LOCKSTEP") ? atoi(getenv("GPFIT_LOCKSTEP")) : 1;
  const bool lockstep = lockstep_env != 0 && !reuse_V;
  if (lockstep) {
    c->lv_valid = false; c->lv32_valid = false;
    GP_HIP(hipEventRecord(c->ev_fork, s));
    GP_HIP(hipStreamWaitEvent(sa, c->ev_fork, 0));
    GP_TRY(launch_pack_lower(V, ldv, n, RP(c->Vbuf), ld, np, sa));
    GP_HIP(hipEventRecord(c->ev_join, sa));
  }
  if (!fork_late && !lockstep) GP_TRY(enqueue_v_chain());

  // ---- main stream: metric, kernel matrix, moments, Cholesky of K~ with its inverse
  GP_TRY(launch_localker<R>(th, c->pix, d, dp, n_rows, n_cols, RP(c->Cmat), dp, nullptr, s));
  GP_TRY(launch_gather(X, ldx, n, c->pix, d, dp, np, RP(c->Xt), ld, RP(c->Xm), dp, s));
  GP_TRY(gemm<R>(s, 1, 1, dp, np, dp, 1.0, RP(c->Cmat), dp, RP(c->Xt), ld, 0.0, RP(c->XCt), ld, 0, 0, 0));
  GP_TRY(launch_qvec(RP(c->Xt), RP(c->XCt), ld, dp, n, np, s0sq, RP(c->Kvec), RP(c->q), s));
  {
    GramArgsT<R> g{};
    g.XCt = RP(c->XCt); g.Xt = RP(c->Xt); g.q1 = RP(c->q); g.q2 = RP(c->q); g.Kout = RP(c->Kbuf); g.Cos = RP(c->Cos);
    g.ld1 = ld; g.ld2 = ld; g.ldk = ld; g.np1 = np; g.np2 = np; g.nv1 = n; g.nv2 = n; g.Kd = dp;
    g.s0sq = s0sq; g.lower = 1; g.pad_identity = 1;
    ProfScope ps(s, (double)np * (np + TILE) * dp, 2);
    GP_TRY(launch_gram(g, s));
  }
  GP_TRY(launch_moments(RP(c->Kvec), RP(c->q), RP(c->Cos), ld, V, ldv, m, r, n, A, lambda0, RP(c->lam_m), RP(c->lam_var), RP(c->fvec),
                        RP(c->wl), c->scal, c->sumA_part, c->info + 2, s));
  phase(1, s);
  // tuning knob: start the V chain only when the K~ chain has factored its leading block of this size
  static const int v_after = getenv("GPFIT_V_AFTER") ? atoi(getenv("GPFIT_V_AFTER")) : 0;
  const bool v_marked = fork_late && !async_call && v_after >= TILE && v_after < np;
  if (lockstep) {
    GP_HIP(hipStreamWaitEvent(s, c->ev_join, 0));   // V is packed
    CholBatchT<R> cb;
    cb.nb = 2;
    cb.A[0] = RP(c->Kbuf); cb.L[0] = RP(c->Lbuf); cb.Li[0] = RP(c->Libuf); cb.Tmp[0] = RP(c->Tmp); cb.info[0] = c->info + 0;
    cb.A[1] = RP(c->Vbuf); cb.L[1] = RP(c->LVbuf); cb.Li[1] = RP(c->LiVbuf); cb.Tmp[1] = RP(c->TmpV); cb.info[1] = c->info + 1;
    cb.ld = ld; cb.ws = 0; cb.sk_ws = c->sk_ws[0]; cb.ctx = c; cb.chain = 0; cb.side_min = side_min;
    GP_TRY(potrf_lockstep<R>(cb, 0, np, 1u, s));
    GP_TRY(launch_logdet(RP(c->LVbuf), ld, n, c->scal + 40, s));
    phase(3, s);
  } else if (fork_late && !v_marked) {
    GP_HIP(hipEventRecord(c->ev_fork, s));
    GP_HIP(hipStreamWaitEvent(sa, c->ev_fork, 0));
    GP_TRY(enqueue_v_chain());
  }
  if (!lockstep) {
    CholBufsT<R> bk{RP(c->Kbuf), RP(c->Lbuf), RP(c->Libuf), RP(c->Tmp), ld, c->info + 0, 0, c->sk_ws[0], c, 0, side_min, half_occ & 2};
    if (v_marked) { bk.mark_ev = c->ev_fork; bk.mark_n = v_after; }
    GP_TRY(potrf_rec<R>(bk, 0, np, true, s));
    if (v_marked) {
      GP_HIP(hipStreamWaitEvent(sa, c->ev_fork, 0));
      GP_TRY(enqueue_v_chain());
    }
  }
  GP_TRY(launch_logdet(RP(c->Lbuf), ld, n, c->scal + 3, s));
  GP_TRY(launch_trmv_lower(RP(c->Libuf), ld, np, RP(c->mpad), RP(c->yv), s));       // y = L^-1 m
  GP_TRY(launch_dot(RP(c->yv), RP(c->yv), np, c->scal + 6, s));                  // m^T K~^-1 m
  GP_TRY(launch_trmv_lower_t(RP(c->Libuf), ld, np, RP(c->yv), RP(c->bv), c->trmv_part, s));  // b = K~^-1 m

  phase(2, s);
  // ---- join: everything that needs both factors
  if (!lockstep) GP_HIP(hipStreamWaitEvent(s, c->ev_join, 0));
  {
    PostJoin<R> pj{RP(c->Libuf), RP(c->LVbuf), RP(c->Cos), RP(c->bv), RP(c->q), RP(c->wl), RP(c->Xm), RP(c->Cmat),
                   RP(c->Tbuf), RP(c->Wbuf), RP(c->Zbuf), RP(c->Tmp), RP(c->Abuf), RP(c->Ybuf), RP(c->tvec),
                   RP(c->Mpart), RP(c->Mmat)};
    if (mixed_grad) {
      // fp64 factorisations, log-determinants and likelihood; fp32 for the N^3-heavy products T, Q, W and the
      // pull-back (T's norm, the trace term of the KL, is therefore fp32-derived: 2e-9 on the loss at N = 8192):
      // single-precision copies of the factors and of the O(N^2) / O(N) operands of the adjoint pass.
      // Li -> Kbuf (its input was destroyed by the factorisation), L_V -> Vbuf (likewise; kept while the
      // V factor is reused), cos(delta) -> TmpV, vectors and the d x d metric into spare buffers.
      auto F = [](double* b) { return reinterpret_cast<float*>(b); };
      const int64_t nn = (int64_t)np * np;
      GP_TRY((launch_reduce_slices<double, float>(c->Libuf, nn, 1, F(c->Kbuf), nn, s)));
      if (!(reuse_V && c->lv32_valid)) GP_TRY((launch_reduce_slices<double, float>(c->LVbuf, nn, 1, F(c->Vbuf), nn, s)));
      c->lv32_valid = true;
      GP_TRY((launch_reduce_slices<double, float>(c->Cos, nn, 1, F(c->TmpV), nn, s)));
      GP_TRY((launch_reduce_slices<double, float>(c->bv, np, 1, F(c->q2), np, s)));
      GP_TRY((launch_reduce_slices<double, float>(c->q, np, 1, F(c->dq1), np, s)));
      GP_TRY((launch_reduce_slices<double, float>(c->wl, np, 1, F(c->dq2), np, s)));
      GP_TRY((launch_reduce_slices<double, float>(c->Xm, (int64_t)np * dp, 1, F(c->Xt2), (int64_t)np * dp, s)));
      GP_TRY((launch_reduce_slices<double, float>(c->Cmat, (int64_t)dp * dp, 1, F(c->dCpad), (int64_t)dp * dp, s)));
      PostJoin<float> pf{F(c->Kbuf), F(c->Vbuf), F(c->TmpV), F(c->q2), F(c->dq1), F(c->dq2), F(c->Xt2), F(c->dCpad),
                         F(c->Tbuf), F(c->Wbuf), F(c->Zbuf), F(c->Tmp), F(c->Abuf), F(c->Ybuf), F(c->tvec),
                         F(c->Mpart), F(c->Mmat)};
      GP_TRY(post_join<float>(c, pf, th, n, np, d, dp, n_rows, n_cols, want_grad, s, phase, !async_call));
    } else {
      GP_TRY(post_join<R>(c, pj, th, n, np, d, dp, n_rows, n_cols, want_grad, s, phase, !async_call));
    }
  }

  if (lam_m_out) GP_HIP(hipMemcpyAsync(lam_m_out, RP(c->lam_m), (size_t)n * sizeof(R), hipMemcpyDeviceToDevice, s));
  if (lam_var_out) GP_HIP(hipMemcpyAsync(lam_var_out, RP(c->lam_var), (size_t)n * sizeof(R), hipMemcpyDeviceToDevice, s));
  if (f_out) GP_HIP(hipMemcpyAsync(f_out, RP(c->fvec), (size_t)n * sizeof(R), hipMemcpyDeviceToDevice, s));
  GP_HIP(hipMemcpyAsync(c->scal_host, c->scal, 64 * sizeof(double), hipMemcpyDeviceToHost, s));
  GP_HIP(hipMemcpyAsync(c->info_host, c->info, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  phase(7, s);
  c->phase_valid = (c->profile == 2 && want_grad && !reuse_V);
  c->last_enqueue_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
  c->pend.active = true; c->pend.stream = s; c->pend.A = A; c->pend.lambda0 = lambda0; c->pend.sigma0 = th.sigma0;
  c->pend.n = n; c->pend.np = np; c->pend.d = d; c->pend.want_grad = want_grad; c->pend.elem_bytes = (int)sizeof(R);
  c->pend.use_done = false;
  if (async_call) return 0;
  return fit_eval_finish(c, out_host);
}

// Several independent units (cells, hyperparameter-grid points: SURVEY 8(e)) of the same N in one call, each on
// its own context (its own workspace), all on the caller's stream.  The factorisations of ALL units -- 2 chains
// per unit -- are one lock-step recursion (potrf_lockstep): its latency-bound leaves and small products are
// shared launches, so their cost is paid once per group instead of once per unit; the products behind the
// factorisations share their launches the same way wherever one unit's product cannot fill the chip
// (post_join_list).  Every unit's numbers are bit-identical to gpfit_fit_eval on its own.
// want_grad bits as gpfit_fit_eval (1 gradients, 2 reuse this context's V factor, 8 mixed precision); bit 2
// (asynchronous) is implied: the call returns after enqueuing and the units are collected one by one with
// gpfit_fit_eval_finish.  rc_out[u]: 0 enqueued (collect it), -2 theta outside the limits (out_host[16 u ..]
// already holds the infinite loss / gradients, nothing to collect).
template <typename R>
static int fit_eval_batch_impl(gpfit_ctx* const* cs, int nu, void* stream, const double* theta6, const double* lower,
                               const double* upper, int n_rows, int n_cols, const R* const* X, int64_t ldx, int64_t N,
                               const R* const* r, const R* const* m, const R* const* V, int64_t ldv, const double* logA,
                               const double* lambda0, int want_grad, double* out_host, int* rc_out) {
  auto RP = [](double* b) { return reinterpret_cast<R*>(b); };
  if (!cs || nu <= 0 || 2 * nu > GEMM_MAXB || !theta6 || !X || !r || !m || !V || !logA || !lambda0 || !out_host || !rc_out ||
      N <= 0) {
    set_error("gpfit_fit_eval_batch: bad argument (1 .. 16 units per call)");
    return -3;
  }
  for (int u = 0; u < nu; ++u) {
    if (!cs[u] || !X[u] || !r[u] || !m[u] || !V[u]) {
      set_error("gpfit_fit_eval_batch: null context or operand");
      return -3;
    }
    if (cs[u]->device != cs[0]->device) {
      set_error("gpfit_fit_eval_batch: the contexts of one call must live on one device");
      return -3;
    }
    for (int v = 0; v < u; ++v)
      if (cs[v] == cs[u]) {
        set_error("gpfit_fit_eval_batch: every unit needs a context of its own");
        return -3;
      }
    if (cs[u]->pend.active) {
      set_error("gpfit_fit_eval_batch: an asynchronous evaluation is pending on one of the contexts");
      return -3;
    }
  }
  DeviceGuard device_guard(cs[0]->device);
  hipStream_t s = (hipStream_t)stream;
  const int n = (int)N, np = (int)round_up(N, TILE);
  const int dfull = n_rows * n_cols;
  const int64_t ld = np;
  const double inf = std::numeric_limits<double>::infinity();
  const bool mixed_grad = (want_grad & 8) != 0 && sizeof(R) == 8 && (want_grad & 1);
  const bool want_reuse = (want_grad & 2) != 0;
  want_grad &= 1;
  const auto t_host0 = std::chrono::steady_clock::now();

  struct Unit { gpfit_ctx* c; int u, d, dp; Theta th; double A; bool reuse_V; };
  Unit un[GEMM_MAXB];
  int na = 0;
  for (int u = 0; u < nu; ++u) {
    gpfit_ctx* c = cs[u];
    const double* theta = theta6 + 6 * u;
    double* out = out_host + 16 * u;
    rc_out[u] = 0;
    if (lower && upper && check_limits(theta, lower, upper) != 0) {   // utils.py:2020-2028
      out[0] = inf;
      out[1] = out[2] = std::numeric_limits<double>::quiet_NaN();
      for (int i = 0; i < 6; ++i) out[3 + i] = inf;
      rc_out[u] = -2;
      continue;
    }
    if (np > c->np_cap || dfull > c->dfull_cap) {
      set_error("gpfit_fit_eval_batch: problem larger than a context's capacity");
      return -3;
    }
    const int d = compute_mask(theta, n_rows, n_cols, nullptr, c->pix_host);
    const int dp = (int)round_up(d, 32);
    if (d <= 0 || dp > c->dp_cap) {
      set_error("gpfit_fit_eval_batch: masked pixel count is zero or exceeds a context's capacity");
      return -3;
    }
    Unit& q = un[na++];
    q.c = c; q.u = u; q.d = d; q.dp = dp; q.th = make_theta(theta); q.A = std::exp(logA[u]);
    q.reuse_V = want_reuse && c->lv_valid && c->lv_n == n && c->lv_bytes == (int)sizeof(R);
    c->cur_n = n; c->cur_np = np; c->cur_d = d; c->cur_dp = dp;
    c->phase_valid = false;
    c->side_ev_next[0] = c->side_ev_next[1] = 0;
  }
  if (na == 0) return 0;
  gpfit_ctx* c0 = un[0].c;
  // tuning aid: GPFIT_BATCH_TIMES=1 prints the three phases of every group (synchronises: not for timed runs)
  static const bool batch_times = getenv("GPFIT_BATCH_TIMES") != nullptr;
  hipEvent_t tev[4] = {nullptr, nullptr, nullptr, nullptr};
  if (batch_times) {
    for (auto& e : tev) (void)hipEventCreate(&e);
    (void)hipEventRecord(tev[0], s);
  }
  // Everything on the caller's stream (post_join_list: a group gets its concurrency from batched launches).
  // ---- phase 1, unit by unit: kernel build, moments, V packed (pixel lists, info words and padded means of all
  // units by one launch)
  {
    GroupPrepT<R> gp{};
    gp.n_units = na; gp.n = n; gp.np = np;
    for (int i = 0; i < na; ++i) {
      gpfit_ctx* c = un[i].c;
      gp.pix_host[i] = c->pix_host; gp.pix[i] = c->pix; gp.d[i] = un[i].d; gp.info[i] = c->info;
      gp.m[i] = m[un[i].u]; gp.mpad[i] = RP(c->mpad);
    }
    GP_TRY(launch_group_prepare(gp, s));
  }
  for (int i = 0; i < na; ++i) {
    Unit& q = un[i];
    gpfit_ctx* c = q.c;
    const int d = q.d, dp = q.dp;
    const double s0sq = q.th.sigma0 * q.th.sigma0;
    g_main_sk_ws = c0->sk_ws[0];
    GP_TRY(launch_localker<R>(q.th, c->pix, d, dp, n_rows, n_cols, RP(c->Cmat), dp, nullptr, s));
    GP_TRY(launch_gather(X[q.u], ldx, n, c->pix, d, dp, np, RP(c->Xt), ld, RP(c->Xm), dp, s));
    GP_TRY(gemm<R>(s, 1, 1, dp, np, dp, 1.0, RP(c->Cmat), dp, RP(c->Xt), ld, 0.0, RP(c->XCt), ld, 0, 0, 0));
    GP_TRY(launch_qvec(RP(c->Xt), RP(c->XCt), ld, dp, n, np, s0sq, RP(c->Kvec), RP(c->q), s));
    {
      GramArgsT<R> g{};
      g.XCt = RP(c->XCt); g.Xt = RP(c->Xt); g.q1 = RP(c->q); g.q2 = RP(c->q); g.Kout = RP(c->Kbuf); g.Cos = RP(c->Cos);
      g.ld1 = ld; g.ld2 = ld; g.ldk = ld; g.np1 = np; g.np2 = np; g.nv1 = n; g.nv2 = n; g.Kd = dp;
      g.s0sq = s0sq; g.lower = 1; g.pad_identity = 1;
      GP_TRY(launch_gram(g, s));
    }
    GP_TRY(launch_moments(RP(c->Kvec), RP(c->q), RP(c->Cos), ld, V[q.u], ldv, m[q.u], r[q.u], n, q.A, lambda0[q.u], RP(c->lam_m),
                          RP(c->lam_var), RP(c->fvec), RP(c->wl), c->scal, c->sumA_part, c->info + 2, s));
    if (!q.reuse_V) {
      c->lv_valid = false; c->lv32_valid = false;
      GP_TRY(launch_pack_lower(V[q.u], ldv, n, RP(c->Vbuf), ld, np, s));
    }
  }
  if (batch_times) (void)hipEventRecord(tev[1], s);
  // ---- phase 2: all factorisations in lock step
  {
    CholBatchT<R> cb;
    uint32_t need = 0;
    for (int i = 0; i < na; ++i) {
      gpfit_ctx* c = un[i].c;
      int b = cb.nb++;
      cb.A[b] = RP(c->Kbuf); cb.L[b] = RP(c->Lbuf); cb.Li[b] = RP(c->Libuf); cb.Tmp[b] = RP(c->Tmp); cb.info[b] = c->info + 0;
      need |= 1u << b;
      if (un[i].reuse_V) continue;
      b = cb.nb++;
      cb.A[b] = RP(c->Vbuf); cb.L[b] = RP(c->LVbuf); cb.Li[b] = RP(c->LiVbuf); cb.Tmp[b] = RP(c->TmpV); cb.info[b] = c->info + 1;
    }
    cb.ld = ld; cb.ws = 0; cb.sk_ws = c0->sk_ws[0]; cb.ctx = nullptr; cb.side_min = 0;
    GP_TRY(potrf_lockstep<R>(cb, 0, np, need, s));
  }
  if (batch_times) (void)hipEventRecord(tev[2], s);
  // ---- phase 3: everything that needs the factors
  auto no_phase = [](int, hipStream_t) {};
  gpfit_ctx* cl[GEMM_MAXB];
  Theta thl[GEMM_MAXB];
  int dl[GEMM_MAXB], dpl[GEMM_MAXB];
  PostJoin<R> pjl[GEMM_MAXB];
  PostJoin<float> pfl[GEMM_MAXB];
  for (int i = 0; i < na; ++i) {
    Unit& q = un[i];
    gpfit_ctx* c = q.c;
    const int dp = q.dp;
    cl[i] = c; thl[i] = q.th; dl[i] = q.d; dpl[i] = dp;
    if (!q.reuse_V) GP_TRY(launch_logdet_pair(RP(c->Lbuf), c->scal + 3, RP(c->LVbuf), c->scal + 40, ld, n, s));
    else GP_TRY(launch_logdet(RP(c->Lbuf), ld, n, c->scal + 3, s));
    GP_TRY(launch_trmv_lower(RP(c->Libuf), ld, np, RP(c->mpad), RP(c->yv), s));
    GP_TRY(launch_dot(RP(c->yv), RP(c->yv), np, c->scal + 6, s));
    GP_TRY(launch_trmv_lower_t(RP(c->Libuf), ld, np, RP(c->yv), RP(c->bv), c->trmv_part, s));
    pjl[i] = PostJoin<R>{RP(c->Libuf), RP(c->LVbuf), RP(c->Cos), RP(c->bv), RP(c->q), RP(c->wl), RP(c->Xm), RP(c->Cmat),
                         RP(c->Tbuf), RP(c->Wbuf), RP(c->Zbuf), RP(c->Tmp), RP(c->Abuf), RP(c->Ybuf), RP(c->tvec),
                         RP(c->Mpart), RP(c->Mmat)};
    if (mixed_grad) {
      auto F = [](double* b) { return reinterpret_cast<float*>(b); };
      const int64_t nn = (int64_t)np * np;
      GP_TRY((launch_reduce_slices<double, float>(c->Libuf, nn, 1, F(c->Kbuf), nn, s)));
      if (!(q.reuse_V && c->lv32_valid)) GP_TRY((launch_reduce_slices<double, float>(c->LVbuf, nn, 1, F(c->Vbuf), nn, s)));
      c->lv32_valid = true;
      GP_TRY((launch_reduce_slices<double, float>(c->Cos, nn, 1, F(c->TmpV), nn, s)));
      GP_TRY((launch_reduce_slices<double, float>(c->bv, np, 1, F(c->q2), np, s)));
      GP_TRY((launch_reduce_slices<double, float>(c->q, np, 1, F(c->dq1), np, s)));
      GP_TRY((launch_reduce_slices<double, float>(c->wl, np, 1, F(c->dq2), np, s)));
      GP_TRY((launch_reduce_slices<double, float>(c->Xm, (int64_t)np * dp, 1, F(c->Xt2), (int64_t)np * dp, s)));
      GP_TRY((launch_reduce_slices<double, float>(c->Cmat, (int64_t)dp * dp, 1, F(c->dCpad), (int64_t)dp * dp, s)));
      pfl[i] = PostJoin<float>{F(c->Kbuf), F(c->Vbuf), F(c->TmpV), F(c->q2), F(c->dq1), F(c->dq2), F(c->Xt2), F(c->dCpad),
                               F(c->Tbuf), F(c->Wbuf), F(c->Zbuf), F(c->Tmp), F(c->Abuf), F(c->Ybuf), F(c->tvec),
                               F(c->Mpart), F(c->Mmat)};
    }
  }
  if (mixed_grad) GP_TRY(post_join_list<float>(na, cl, pfl, thl, n, np, dl, dpl, n_rows, n_cols, want_grad, s, no_phase, false));
  else GP_TRY(post_join_list<R>(na, cl, pjl, thl, n, np, dl, dpl, n_rows, n_cols, want_grad, s, no_phase, false));
  {
    GroupCollectT gc{};
    gc.n_units = na;
    for (int i = 0; i < na; ++i) {
      gpfit_ctx* c = un[i].c;
      gc.scal[i] = c->scal; gc.info[i] = c->info; gc.scal_host[i] = c->scal_host; gc.info_host[i] = c->info_host;
    }
    GP_TRY(launch_group_collect(gc, s));
  }
  // one completion event for the group, recorded on the leader's context and waited on by every unit's finish
  if (!c0->pend.done) GP_HIP(hipEventCreateWithFlags(&c0->pend.done, hipEventDisableTiming));
  GP_HIP(hipEventRecord(c0->pend.done, s));
  for (int i = 0; i < na; ++i) {
    Unit& q = un[i];
    gpfit_ctx* c = q.c;
    if (c != c0) {
      // (a context's own event so that it can outlive the leader: recording is a barrier packet, no kernel)
      if (!c->pend.done) GP_HIP(hipEventCreateWithFlags(&c->pend.done, hipEventDisableTiming));
      GP_HIP(hipEventRecord(c->pend.done, s));
    }
    c->pend.use_done = true;
    c->pend.active = true; c->pend.stream = s; c->pend.A = q.A; c->pend.lambda0 = lambda0[q.u]; c->pend.sigma0 = q.th.sigma0;
    c->pend.n = n; c->pend.np = np; c->pend.d = q.d; c->pend.want_grad = want_grad; c->pend.elem_bytes = (int)sizeof(R);
  }
  c0->last_enqueue_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
  if (batch_times) {
    (void)hipEventRecord(tev[3], s);
    (void)hipEventSynchronize(tev[3]);
    float a = 0, b = 0, d3 = 0;
    (void)hipEventElapsedTime(&a, tev[0], tev[1]);
    (void)hipEventElapsedTime(&b, tev[1], tev[2]);
    (void)hipEventElapsedTime(&d3, tev[2], tev[3]);
    fprintf(stderr, "[gpfit batch] %d units: build %.3f ms, lock-step factorisations %.3f ms, post %.3f ms (enqueue %.2f ms)\n", na, a,
            b, d3, c0->last_enqueue_ms);
    for (auto& e : tev) (void)hipEventDestroy(e);
  }
  return 0;
}

// Gradient pull-back for an externally supplied adjoint: out6[p] = sum_ij W_ij dK~_p,ij +
// sum_i gvec_i dKvec_p,i with the reference's analytic dK~_p / dKvec_p (utils.py:996-1021,
// 1036-1044) at theta, WITHOUT materialising any dK: the same contraction to the d x d metric the
// fused full-rank unit uses.  This is what the truncated-rank (B-projected) closure needs once its
// n x n adjoints have been lifted to W = (B G_Kb~ + G_Kb) B^T (utils._closure_projected).
static int grad_pullback_impl(gpfit_ctx* c, void* stream, const double* theta, int n_rows, int n_cols, const double* X,
                              int64_t ldx, int64_t N, const double* W, int64_t ldw, const double* gvec,
                              double* out6) {
  using R = double;
  auto RP = [](double* b) { return b; };
  if (!c || !theta || !X || !W || !gvec || !out6 || N <= 0) {
    set_error("gpfit_grad_pullback: bad argument");
    return -3;
  }
  GP_CTX_ENTER(c, "gpfit_grad_pullback");
  hipStream_t s = (hipStream_t)stream;
  const int n = (int)N, np = (int)round_up(N, TILE);
  const int dfull = n_rows * n_cols;
  if (np > c->np_cap || dfull > c->dfull_cap) {
    set_error("gpfit_grad_pullback: problem larger than the context capacity");
    return -3;
  }
  const int d = compute_mask(theta, n_rows, n_cols, nullptr, c->pix_host);
  const int dp = (int)round_up(d, 32);
  if (d <= 0 || dp > c->dp_cap) {
    set_error("gpfit_grad_pullback: masked pixel count is zero or exceeds the context capacity");
    return -3;
  }
  const Theta th = make_theta(theta);
  const double s0sq = th.sigma0 * th.sigma0;
  const int64_t ld = np;
  c->lv_valid = false; c->lv32_valid = false;  // the workspace matrices are reused
  g_main_sk_ws = c->sk_ws[0];
  GP_HIP(hipMemcpyAsync(c->pix, c->pix_host, (size_t)d * sizeof(int), hipMemcpyHostToDevice, s));
  GP_TRY(launch_localker<R>(th, c->pix, d, dp, n_rows, n_cols, RP(c->Cmat), dp, nullptr, s));
  GP_TRY(launch_gather(X, ldx, n, c->pix, d, dp, np, RP(c->Xt), ld, RP(c->Xm), dp, s));
  GP_TRY(gemm<R>(s, 1, 1, dp, np, dp, 1.0, RP(c->Cmat), dp, RP(c->Xt), ld, 0.0, RP(c->XCt), ld, 0, 0, 0));
  GP_TRY(launch_qvec(RP(c->Xt), RP(c->XCt), ld, dp, n, np, s0sq, RP(c->Kvec), RP(c->q), s));
  {
    GramArgsT<R> g{};
    g.XCt = RP(c->XCt); g.Xt = RP(c->Xt); g.q1 = RP(c->q); g.q2 = RP(c->q); g.Kout = RP(c->Kbuf); g.Cos = RP(c->Cos);
    g.ld1 = ld; g.ld2 = ld; g.ldk = ld; g.np1 = np; g.np2 = np; g.nv1 = n; g.nv2 = n; g.Kd = dp;
    g.s0sq = s0sq; g.lower = 1; g.pad_identity = 1;
    GP_TRY(launch_gram(g, s));
  }
  GP_TRY(launch_pack_lower(W, ldw, n, RP(c->Wbuf), ld, np, s));
  GP_HIP(hipMemsetAsync(RP(c->bv), 0, (size_t)np * sizeof(R), s));             // no -1/2 b b^T term here
  GP_HIP(hipMemsetAsync(RP(c->wl), 0, (size_t)np * sizeof(R), s));
  GP_TRY(launch_scale_copy<R>(RP(c->wl), gvec, n, -1.0, s));                    // t_i = u_i / q_i + gvec_i
  GP_TRY(launch_adjoint(RP(c->Wbuf), RP(c->Cos), ld, RP(c->bv), RP(c->q), n, np, RP(c->Abuf), c->upart, c->vpart,
                        c->sumA_part, s));
  const int t64 = np / 64;
  GP_TRY(launch_adjoint_reduce(c->upart, c->vpart, c->sumA_part, t64, t64 * (t64 + 1) / 2, RP(c->q), RP(c->wl), n, np,
                               RP(c->tvec), c->rpad, c->scal + 7, s));
  GP_TRY(gemm<R>(s, 1, 1, np, dp, np, 1.0, RP(c->Abuf), ld, RP(c->Xm), dp, 0.0, RP(c->Ybuf), dp, 0, 0, 0));
  GP_TRY(launch_rowscale_add(RP(c->Ybuf), dp, RP(c->Xm), dp, RP(c->tvec), np, dp, s));
  {
    GemmArgsT<R> g{};
    g.A = RP(c->Xm); g.B = RP(c->Ybuf); g.C = RP(c->Mpart);
    g.lda = dp; g.ldb = dp; g.ldc = dp;
    g.M = dp; g.N = dp; g.K = np;
    g.alpha = 1.0; g.beta = 0.0; g.a_kmajor = 1; g.b_kmajor = 1;
    g.batch = 1; g.split_k = c->split_k_M; g.sC = (int64_t)dp * dp;
    GP_TRY(launch_gemm(g, s));
    GP_TRY(launch_reduce_slices(RP(c->Mpart), (int64_t)dp * dp, c->split_k_M, RP(c->Mmat), (int64_t)dp * dp, s));
  }
  GP_TRY(launch_metric_contract(th, c->pix, d, n_rows, n_cols, RP(c->Cmat), dp, RP(c->Mmat), dp, c->scal + 10, c->upart, c->info + 3, s));
  GP_HIP(hipMemcpyAsync(c->scal_host, c->scal, 64 * sizeof(double), hipMemcpyDeviceToHost, s));
  GP_HIP(hipStreamSynchronize(s));
  const double* sc = c->scal_host;
  out6[0] = th.sigma0 * (2.0 * sc[9] + 2.0 * sc[7]) - 2.0 * th.sigma0 * sc[8];  // sigma_0
  out6[1] = sc[13];  // eps_0x
  out6[2] = sc[14];  // eps_0y
  out6[3] = sc[11];  // -2log2beta
  out6[4] = sc[12];  // -log2rho2
  out6[5] = sc[10];  // Amp
  return 0;
}

// Truncated-rank (B-projected) M-step closure with the inducing set = the training set
// (utils.py:2030-2099 with n < n_tilde = n_t), fused: kernel build, projection on the kept
// eigen-directions B, Cholesky of the n x n matrices, moments / likelihood / KL, the n x n and N x n
// adjoints of the loss, their lift W = (B G_K~b + G_Kb) B^T and the pull-back to the metric, all on
// the device in one call (the algebra of utils._closure_projected).  Every
// N x n matrix lives zero-padded to nb = ceil(n / 128) 128 columns in one of the context's N x N
// work matrices; the n x n ones carry the identity on their padding (log-determinants and solves
// are unaffected, the padding of G_K~b cancels to zero).
static int fit_eval_projected_impl(gpfit_ctx* c, void* stream, const double* theta, const double* lower,
                                   const double* upper, int n_rows, int n_cols, const double* X, int64_t ldx, int64_t N,
                                   const double* r, const double* B, int64_t ldb, int64_t n_kept, const double* m_b,
                                   const double* V_b, int64_t ldvb, double logA, double lambda0, double* out_host) {
  using R = double;
  if (!c || !theta || !X || !r || !B || !m_b || !V_b || !out_host || N <= 0 || n_kept <= 0 || n_kept > N) {
    set_error("gpfit_fit_eval_projected: bad argument");
    return -3;
  }
  GP_CTX_ENTER(c, "gpfit_fit_eval_projected");
  const double inf = std::numeric_limits<double>::infinity();
  if (lower && upper && check_limits(theta, lower, upper) != 0) {
    out_host[0] = inf;
    out_host[1] = out_host[2] = std::numeric_limits<double>::quiet_NaN();
    for (int i = 0; i < 6; ++i) out_host[3 + i] = inf;
    return -2;
  }
  hipStream_t s = (hipStream_t)stream;
  const int n = (int)N, np = (int)round_up(N, TILE), nk = (int)n_kept, nb = (int)round_up(n_kept, TILE);
  const int dfull = n_rows * n_cols;
  if (np > c->np_cap || dfull > c->dfull_cap) {
    set_error("gpfit_fit_eval_projected: problem larger than the context capacity");
    return -3;
  }
  const int d = compute_mask(theta, n_rows, n_cols, nullptr, c->pix_host);
  const int dp = (int)round_up(d, 32);
  if (d <= 0 || dp > c->dp_cap) {
    set_error("gpfit_fit_eval_projected: masked pixel count is zero or exceeds the context capacity");
    return -3;
  }
  const Theta th = make_theta(theta);
  const double s0sq = th.sigma0 * th.sigma0, A = std::exp(logA);
  const int64_t ld = np, lb = nb;
  c->lv_valid = false; c->lv32_valid = false;
  c->side_ev_next[0] = c->side_ev_next[1] = 0;
  g_main_sk_ws = c->sk_ws[0];
  ++g_eval_count;
  prof_begin(c);
  struct ProfGuard { gpfit_ctx* c; ~ProfGuard() { prof_end(c); } } prof_guard{c};
  double *Kt = c->Kbuf, *Bp = c->Lbuf, *Kb = c->Libuf, *aV = c->Tbuf, *Ga = c->Zbuf, *GaKi = c->Tmp, *W = c->Wbuf;
  double *S1 = c->Vbuf, *S2 = c->LVbuf, *S3 = c->LiVbuf, *S4 = c->TmpV;   // n x n scratch (leading dimension nb)
  double *mbp = c->mpad, *bvec = c->yv, *gm = c->dq1, *gv = c->dq2;
  GP_HIP(hipMemsetAsync(c->info, 0, 4 * sizeof(int), s));
  GP_HIP(hipMemcpyAsync(c->pix, c->pix_host, (size_t)d * sizeof(int), hipMemcpyHostToDevice, s));
  // (log|V_b| of :1326: V_b is factored together with K~_b below -- one lock-step recursion on this stream, in
  // four work matrices nothing else needs before the adjoints: Abuf, Wbuf, Zbuf, Tmp)
  // ---- kernel build (as the full-rank unit): C, X masked, K~ (lower tiles -> mirrored), cos, Kvec, q
  GP_TRY(launch_localker<R>(th, c->pix, d, dp, n_rows, n_cols, c->Cmat, dp, nullptr, s));
  GP_TRY(launch_gather(X, ldx, n, c->pix, d, dp, np, c->Xt, ld, c->Xm, dp, s));
  GP_TRY(gemm<R>(s, 1, 1, dp, np, dp, 1.0, c->Cmat, dp, c->Xt, ld, 0.0, c->XCt, ld, 0, 0, 0));
  GP_TRY(launch_qvec(c->Xt, c->XCt, ld, dp, n, np, s0sq, c->Kvec, c->q, s));
  {
    GramArgsT<R> g{};
    g.XCt = c->XCt; g.Xt = c->Xt; g.q1 = c->q; g.q2 = c->q; g.Kout = Kt; g.Cos = c->Cos;
    g.ld1 = ld; g.ld2 = ld; g.ldk = ld; g.np1 = np; g.np2 = np; g.nv1 = n; g.nv2 = n; g.Kd = dp;
    g.s0sq = s0sq; g.lower = 1; g.pad_identity = 1;
    g.mirror = 1;   // K~ is multiplied from the left below: stored in full by the tiles themselves
    ProfScope ps(s, (double)np * (np + TILE) * dp, 2);
    GP_TRY(launch_gram(g, s));
  }
  // ---- projection (utils.py:2047-2049): K_b = K~ B, K~_b = sym(B^T K_b)
  GP_TRY(launch_pad_copy(B, ldb, n, nk, Bp, lb, np, nb, s));
  GP_HIP(hipMemsetAsync(mbp, 0, (size_t)np * sizeof(double), s));
  GP_HIP(hipMemcpyAsync(mbp, m_b, (size_t)nk * sizeof(double), hipMemcpyDeviceToDevice, s));
  GP_TRY(gemm_splitk<R>(s, 0, 1, np, nb, np, 1.0, Kt, ld, Bp, lb, Kb, lb, splitk_for(np, nb, np), c->Wbuf, (int64_t)c->np_cap * c->np_cap));
  GP_TRY(gemm_splitk<R>(s, 1, 1, nb, nb, np, 1.0, Bp, lb, Kb, lb, S4, lb, splitk_for(nb, nb, np), c->Wbuf, (int64_t)c->np_cap * c->np_cap));
  GP_TRY(launch_symmetrize_avg(S4, lb, nk, s));                                             // :2048
  GP_TRY(launch_pack_lower(S4, lb, nk, S1, lb, nb, s));
  GP_TRY(launch_pack_lower(V_b, ldvb, nk, c->Abuf, lb, nb, s));
  {
    // K~_b = L L^T with L^-1 (:2067) and V_b = L_V L_V^T (log|V_b|, :1326) in lock step
    CholBatchT<R> cb;
    cb.nb = 2;
    cb.A[0] = S1; cb.L[0] = S2; cb.Li[0] = S3; cb.Tmp[0] = S4; cb.info[0] = c->info + 0;
    cb.A[1] = c->Abuf; cb.L[1] = c->Wbuf; cb.Li[1] = c->Zbuf; cb.Tmp[1] = c->Tmp; cb.info[1] = c->info + 1;
    cb.ld = lb; cb.ws = 0; cb.sk_ws = c->sk_ws[0]; cb.ctx = nullptr; cb.side_min = 0;
    GP_TRY(potrf_lockstep<R>(cb, 0, nb, 1u, s));
  }
  GP_TRY(launch_logdet(c->Wbuf, lb, nk, c->scal + 40, s));
  GP_TRY(launch_logdet(S2, lb, nk, c->scal + 3, s));
  // K~_b^-1 = L^-T L^-1 (lower tiles, mirrored)
  GP_TRY(gemm<R>(s, 1, 1, nb, nb, nb, 1.0, S3, lb, S3, lb, 0.0, S1, lb, 1, 2, 1));
  GP_TRY(launch_symmetrize(S1, lb, nb, s));
  double* Ki = S1;
  // V_b padded (identity on the padding), a V = B V_b, K~_b^-1 V_b and its trace, K~_b^-1 V_b K~_b^-1
  GP_TRY(launch_pack_lower(V_b, ldvb, nk, S2, lb, nb, s));
  GP_TRY(launch_symmetrize(S2, lb, nb, s));
  GP_TRY(gemm<R>(s, 0, 1, np, nb, nb, 1.0, Bp, lb, S2, lb, 0.0, aV, lb, 0, 0, 0));
  GP_TRY(gemm<R>(s, 0, 1, nb, nb, nb, 1.0, Ki, lb, S2, lb, 0.0, S3, lb, 0, 0, 0));
  GP_TRY(launch_proj_trace(S3, lb, nk, c->scal + 5, s));                                     // tr(K~_b^-1 V_b)
  GP_TRY(gemm<R>(s, 0, 1, nb, nb, nb, 1.0, S3, lb, Ki, lb, 0.0, S4, lb, 0, 0, 0));            // P1
  GP_TRY(launch_symv_lower(Ki, lb, nb, mbp, bvec, s));                                        // b = K~_b^-1 m_b
  GP_TRY(launch_dot(mbp, bvec, nb, c->scal + 6, s));
  // ---- moments, rate, likelihood pieces (:1090, 1101, 1138, 1243) and the per-point adjoints
  GP_TRY(launch_proj_moments(Bp, Kb, aV, lb, nb, mbp, c->Kvec, r, n, A, lambda0, c->lam_m, c->lam_var, c->fvec, gm, gv,
                             c->upart, c->scal + 0, s));
  // ---- adjoints (utils._closure_projected): G_a, G_Kb, G_K~b, W
  GP_TRY(launch_proj_ga(Kb, aV, lb, nb, n, np, gm, gv, mbp, Ga, s));
  GP_TRY(gemm<R>(s, 0, 1, np, nb, nb, 1.0, Ga, lb, Ki, lb, 0.0, GaKi, lb, 0, 0, 0));
  GP_TRY(gemm_splitk<R>(s, 1, 1, nb, nb, np, 1.0, Bp, lb, GaKi, lb, c->Abuf, lb, splitk_for(nb, nb, np), c->Wbuf, (int64_t)c->np_cap * c->np_cap));   // P2 = B^T G_a K~_b^-1
  GP_TRY(launch_proj_gktb(Ki, S4, c->Abuf, lb, nb, bvec, S3, s));                            // G_K~b
  GP_TRY(launch_proj_gkb(Bp, lb, nb, n, np, gv, GaKi, s));                                   // G_Kb (in place)
  GP_TRY(gemm<R>(s, 0, 1, np, nb, nb, 1.0, Bp, lb, S3, lb, 1.0, GaKi, lb, 0, 0, 0));          // + B G_K~b
  // W = sym((.) B^T).  With P = B G_K~b + G_Kb:  1/2 (P B^T + B P^T) = 1/2 [P | B] [B | P]^T -- ONE product with
  // k = 2 nb that writes the lower tiles only (what the adjoint pass reads): the same flops as the full P B^T, and
  // neither its upper half nor the averaging pass over N x N exist.  (Falls back to the two steps when 2 nb
  // columns do not fit the N x N scratch matrices, i.e. when hardly anything was truncated.)
  if (2 * (int64_t)nb * np <= (int64_t)c->np_cap * c->np_cap) {
    double *Cat1 = Ga, *Cat2 = aV;   // G_a and a V_b are dead
    const int64_t l2b = 2 * lb;
    GP_TRY(launch_pad_copy(GaKi, lb, np, nb, Cat1, l2b, np, nb, s));
    GP_TRY(launch_pad_copy(Bp, lb, np, nb, Cat1 + nb, l2b, np, nb, s));
    GP_TRY(launch_pad_copy(Bp, lb, np, nb, Cat2, l2b, np, nb, s));
    GP_TRY(launch_pad_copy(GaKi, lb, np, nb, Cat2 + nb, l2b, np, nb, s));
    GP_TRY(gemm<R>(s, 0, 0, np, np, 2 * nb, 0.5, Cat1, l2b, Cat2, l2b, 0.0, W, ld, 1, 0, 0));
  } else {
    GP_TRY(gemm<R>(s, 0, 0, np, np, nb, 1.0, GaKi, lb, Bp, lb, 0.0, W, ld, 0, 0, 0));
    GP_TRY(launch_symmetrize_avg(W, ld, np, s));
  }
  // ---- pull-back of <W, dK~_p> + <gvec, dKvec_p> to the metric (as gpfit_grad_pullback; gvec = -g_v)
  GP_HIP(hipMemsetAsync(c->bv, 0, (size_t)np * sizeof(R), s));
  GP_HIP(hipMemsetAsync(c->wl, 0, (size_t)np * sizeof(R), s));
  GP_TRY(launch_scale_copy<R>(c->wl, gv, n, 1.0, s));
  GP_TRY(launch_adjoint(W, c->Cos, ld, c->bv, c->q, n, np, c->Abuf, c->upart, c->vpart, c->sumA_part, s));
  const int t64 = np / 64;
  GP_TRY(launch_adjoint_reduce(c->upart, c->vpart, c->sumA_part, t64, t64 * (t64 + 1) / 2, c->q, c->wl, n, np, c->tvec,
                               c->rpad, c->scal + 7, s));
  GP_TRY(gemm<R>(s, 1, 1, np, dp, np, 1.0, c->Abuf, ld, c->Xm, dp, 0.0, c->Ybuf, dp, 0, 0, 0));
  GP_TRY(launch_rowscale_add(c->Ybuf, dp, c->Xm, dp, c->tvec, np, dp, s));
  {
    GemmArgsT<R> g{};
    g.A = c->Xm; g.B = c->Ybuf; g.C = c->Mpart;
    g.lda = dp; g.ldb = dp; g.ldc = dp;
    g.M = dp; g.N = dp; g.K = np;
    g.alpha = 1.0; g.beta = 0.0; g.a_kmajor = 1; g.b_kmajor = 1;
    g.batch = 1; g.split_k = c->split_k_M; g.sC = (int64_t)dp * dp;
    GP_TRY(launch_gemm(g, s));
    GP_TRY(launch_reduce_slices(c->Mpart, (int64_t)dp * dp, c->split_k_M, c->Mmat, (int64_t)dp * dp, s));
  }
  GP_TRY(launch_metric_contract(th, c->pix, d, n_rows, n_cols, c->Cmat, dp, c->Mmat, dp, c->scal + 10, c->upart, c->info + 3, s));
  GP_HIP(hipMemcpyAsync(c->scal_host, c->scal, 64 * sizeof(double), hipMemcpyDeviceToHost, s));
  GP_HIP(hipMemcpyAsync(c->info_host, c->info, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  GP_HIP(hipStreamSynchronize(s));
  const double* sc = c->scal_host;
  const double loglik = A * sc[0] + lambda0 * sc[1] - sc[2];                                 // :1243
  const double KL = -0.5 * sc[40] + 0.5 * sc[3] + 0.5 * sc[6] + 0.5 * sc[5];                  // :1326
  out_host[0] = -(loglik - KL);
  out_host[1] = loglik;
  out_host[2] = KL;
  out_host[3] = th.sigma0 * (2.0 * sc[9] + 2.0 * sc[7]) - 2.0 * th.sigma0 * sc[8];
  out_host[4] = sc[13];
  out_host[5] = sc[14];
  out_host[6] = sc[11];
  out_host[7] = sc[12];
  out_host[8] = sc[10];
  out_host[9] = sc[3];
  out_host[10] = sc[40];
  out_host[11] = sc[5];
  out_host[12] = sc[6];
  out_host[13] = (double)d;
  out_host[14] = (double)c->info_host[0];
  out_host[15] = (double)c->info_host[1];
  if (c->info_host[0] != 0) {
    set_error("gpfit_fit_eval_projected: Cholesky of the projected K_tilde failed (non-positive pivot)");
    return c->info_host[0];
  }
  if (c->info_host[1] != 0) {
    set_error("gpfit_fit_eval_projected: Cholesky of V_b failed (non-positive pivot)");
    return c->info_host[1];
  }
  return 0;
}

// Sparse M-step closure (n_tilde < n_t: K[n_t][n_tilde] != K~, a = K_b K~_b^-1 with non-zero da_p;
// utils.py:2030-2099, 1114-1120), fused like the truncated-rank one above.  Two kernel objects (the
// square K~ on the inducing stimuli and the rectangular K between training and inducing stimuli), the
// same n x n algebra with a = K_b K~_b^-1 in place of B, and two adjoints to pull back:
//   W~  = sym(B G_K~b B^T)   against dK~_p   (square pull-back on the inducing stimuli)
//   W_K = G_Kb B^T           against dK_p    (rectangular pull-back, with the dKvec term riding on it)
// (algebra of utils._closure_sparse).  The two d x d matrices are added before the one contraction
// with dC_p, which is linear in them.
static int fit_eval_sparse_impl(gpfit_ctx* c, void* stream, const double* theta, const double* lower,
                                const double* upper, int n_rows, int n_cols, const double* X, int64_t ldx, int64_t N,
                                const double* Xtilde, int64_t ldxt, int64_t Ntilde, const double* r, const double* B,
                                int64_t ldb, int64_t n_kept, const double* m_b, const double* V_b, int64_t ldvb,
                                double logA, double lambda0, double* out_host) {
  using R = double;
  if (!c || !theta || !X || !Xtilde || !r || !B || !m_b || !V_b || !out_host || N <= 0 || Ntilde <= 0 || n_kept <= 0 ||
      n_kept > Ntilde) {
    set_error("gpfit_fit_eval_sparse: bad argument");
    return -3;
  }
  GP_CTX_ENTER(c, "gpfit_fit_eval_sparse");
  const double inf = std::numeric_limits<double>::infinity();
  if (lower && upper && check_limits(theta, lower, upper) != 0) {
    out_host[0] = inf;
    out_host[1] = out_host[2] = std::numeric_limits<double>::quiet_NaN();
    for (int i = 0; i < 6; ++i) out_host[3 + i] = inf;
    return -2;
  }
  hipStream_t s = (hipStream_t)stream;
  const int n1 = (int)N, n2 = (int)Ntilde, nk = (int)n_kept;
  const int np1 = (int)round_up(N, TILE), np2 = (int)round_up(Ntilde, TILE), nb = (int)round_up(n_kept, TILE);
  const int dfull = n_rows * n_cols;
  if (np1 > c->np_cap || np2 > c->np_cap || dfull > c->dfull_cap) {
    set_error("gpfit_fit_eval_sparse: problem larger than the context capacity");
    return -3;
  }
  const int d = compute_mask(theta, n_rows, n_cols, nullptr, c->pix_host);
  const int dp = (int)round_up(d, 32);
  if (d <= 0 || dp > c->dp_cap) {
    set_error("gpfit_fit_eval_sparse: masked pixel count is zero or exceeds the context capacity");
    return -3;
  }
  const Theta th = make_theta(theta);
  const double s0sq = th.sigma0 * th.sigma0, A = std::exp(logA);
  const int64_t l2 = np2, lb = nb;
  c->lv_valid = false; c->lv32_valid = false;
  c->side_ev_next[0] = c->side_ev_next[1] = 0;
  g_main_sk_ws = c->sk_ws[0];
  ++g_eval_count;
  prof_begin(c);
  struct ProfGuard { gpfit_ctx* c; ~ProfGuard() { prof_end(c); } } prof_guard{c};
  double *X1m = c->Xm, *X2m = c->XDt, *Zm = c->XDt2;
  double *Kt = c->Kbuf, *CosT = c->Cos, *Kr = c->Lbuf, *CosR = c->Libuf, *Bp = c->Tbuf, *Kb = c->Zbuf, *am = c->Tmp,
         *aV = c->Abuf;
  double *S1 = c->Vbuf, *S2 = c->LVbuf, *S3 = c->LiVbuf, *S4 = c->TmpV;
  double *mbp = c->mpad, *bvec = c->yv, *gm = c->dq1, *gv = c->dq2, *gvec = c->hvec;
  GP_HIP(hipMemsetAsync(c->info, 0, 4 * sizeof(int), s));
  GP_HIP(hipMemcpyAsync(c->pix, c->pix_host, (size_t)d * sizeof(int), hipMemcpyHostToDevice, s));
  // ---- kernel objects: C; training side (x: Xt, XCt, q, Kvec), inducing side (xtilde: Xt2, XCt2, q2)
  GP_TRY(launch_localker<R>(th, c->pix, d, dp, n_rows, n_cols, c->Cmat, dp, nullptr, s));
  GP_TRY(launch_gather(X, ldx, n1, c->pix, d, dp, np1, c->Xt, (int64_t)np1, X1m, dp, s));
  GP_TRY(gemm<R>(s, 1, 1, dp, np1, dp, 1.0, c->Cmat, dp, c->Xt, np1, 0.0, c->XCt, np1, 0, 0, 0));
  GP_TRY(launch_qvec(c->Xt, c->XCt, np1, dp, n1, np1, s0sq, c->Kvec, c->q, s));
  GP_TRY(launch_gather(Xtilde, ldxt, n2, c->pix, d, dp, np2, c->Xt2, l2, X2m, dp, s));
  GP_TRY(gemm<R>(s, 1, 1, dp, np2, dp, 1.0, c->Cmat, dp, c->Xt2, l2, 0.0, c->XCt2, l2, 0, 0, 0));
  GP_TRY(launch_qvec(c->Xt2, c->XCt2, l2, dp, n2, np2, s0sq, c->hvec, c->q2, s));
  {
    GramArgsT<R> g{};  // K~ = acosker(xtilde, xtilde): lower tiles, mirrored below
    g.XCt = c->XCt2; g.Xt = c->Xt2; g.q1 = c->q2; g.q2 = c->q2; g.Kout = Kt; g.Cos = CosT;
    g.ld1 = l2; g.ld2 = l2; g.ldk = l2; g.np1 = np2; g.np2 = np2; g.nv1 = n2; g.nv2 = n2; g.Kd = dp;
    g.s0sq = s0sq; g.lower = 1; g.pad_identity = 1;
    g.mirror = 1;
    ProfScope ps(s, (double)np2 * (np2 + TILE) * dp, 2);
    GP_TRY(launch_gram(g, s));
  }
  {
    GramArgsT<R> g{};  // K = acosker(x, xtilde): rectangular, with its cosine matrix
    g.XCt = c->XCt; g.Xt = c->Xt2; g.q1 = c->q; g.q2 = c->q2; g.Kout = Kr; g.Cos = CosR;
    g.ld1 = np1; g.ld2 = l2; g.ldk = l2; g.np1 = np1; g.np2 = np2; g.nv1 = n1; g.nv2 = n2; g.Kd = dp;
    g.s0sq = s0sq; g.lower = 0; g.pad_identity = 0;
    g.ldcos = l2;
    ProfScope ps(s, 2.0 * np1 * np2 * dp, 2);
    GP_TRY(launch_gram(g, s));
  }
  // ---- projection (:2047-2049, 2067-2068): K_b = K B, K~_b = sym(B^T K~ B), a = K_b K~_b^-1
  GP_TRY(launch_pad_copy(B, ldb, n2, nk, Bp, lb, np2, nb, s));
  GP_HIP(hipMemsetAsync(mbp, 0, (size_t)c->np_cap * sizeof(double), s));
  GP_HIP(hipMemcpyAsync(mbp, m_b, (size_t)nk * sizeof(double), hipMemcpyDeviceToDevice, s));
  // (skinny products with a long k are cut into k slabs, gemm_splitk; Wbuf is free until the adjoints)
  GP_TRY(gemm_splitk<R>(s, 0, 1, np1, nb, np2, 1.0, Kr, l2, Bp, lb, Kb, lb, splitk_for(np1, nb, np2), c->Wbuf, (int64_t)c->np_cap * c->np_cap));
  GP_TRY(gemm_splitk<R>(s, 0, 1, np2, nb, np2, 1.0, Kt, l2, Bp, lb, am, lb, splitk_for(np2, nb, np2), c->Wbuf, (int64_t)c->np_cap * c->np_cap));   // K~ B (temporary)
  GP_TRY(gemm_splitk<R>(s, 1, 1, nb, nb, np2, 1.0, Bp, lb, am, lb, S4, lb, splitk_for(nb, nb, np2), c->Wbuf, (int64_t)c->np_cap * c->np_cap));
  GP_TRY(launch_symmetrize_avg(S4, lb, nk, s));
  GP_TRY(launch_pack_lower(S4, lb, nk, S1, lb, nb, s));
  {
    // K~_b = L L^T with L^-1 and V_b = L_V L_V^T (log|V_b|, :1326) in lock step.  The V_b chain takes four work
    // matrices that are dead between the projections above and the adjoints below, each of the context's full
    // np_cap^2 size (nb <= np2 <= np_cap: an nb x nb chain fits whatever n_kept is): Wbuf (free until P2), Kbuf
    // and Lbuf (K~ and K are consumed by the projections; rewritten as G_a and G_a K~_b^-1 further down) and Abuf
    // (a V_b is formed behind the chain).  The cosine matrices in Cos / Libuf stay untouched.
    double *Va = c->Wbuf, *Vl = c->Kbuf, *Vli = c->Lbuf, *Vt = c->Abuf;
    const int64_t slot = (int64_t)nb * nb;
    GP_TRY(launch_pack_lower(V_b, ldvb, nk, Va, lb, nb, s));
    CholBatchT<R> cb;
    cb.nb = 2;
    cb.A[0] = S1; cb.L[0] = S2; cb.Li[0] = S3; cb.Tmp[0] = S4; cb.info[0] = c->info + 0;
    cb.A[1] = Va; cb.L[1] = Vl; cb.Li[1] = Vli; cb.Tmp[1] = Vt; cb.info[1] = c->info + 1;
    cb.ld = lb; cb.ws = 0; cb.sk_ws = c->sk_ws[0]; cb.ctx = nullptr; cb.side_min = 0;
    for (double* p : {Vl, Vli, Vt})   // tiles above the diagonal read as zero
      GP_HIP(hipMemsetAsync(p, 0, (size_t)slot * sizeof(double), s));
    GP_TRY(potrf_lockstep<R>(cb, 0, nb, 1u, s));
    GP_TRY(launch_logdet(Vl, lb, nk, c->scal + 40, s));
  }
  GP_TRY(launch_logdet(S2, lb, nk, c->scal + 3, s));
  GP_TRY(gemm<R>(s, 1, 1, nb, nb, nb, 1.0, S3, lb, S3, lb, 0.0, S1, lb, 1, 2, 1));
  GP_TRY(launch_symmetrize(S1, lb, nb, s));
  double* Ki = S1;
  GP_TRY(launch_pack_lower(V_b, ldvb, nk, S2, lb, nb, s));
  GP_TRY(launch_symmetrize(S2, lb, nb, s));
  GP_TRY(gemm<R>(s, 0, 1, nb, nb, nb, 1.0, Ki, lb, S2, lb, 0.0, S3, lb, 0, 0, 0));            // K~_b^-1 V_b
  GP_TRY(launch_proj_trace(S3, lb, nk, c->scal + 5, s));
  GP_TRY(gemm<R>(s, 0, 1, nb, nb, nb, 1.0, S3, lb, Ki, lb, 0.0, S4, lb, 0, 0, 0));            // P1
  GP_TRY(gemm<R>(s, 0, 1, np1, nb, nb, 1.0, Kb, lb, Ki, lb, 0.0, am, lb, 0, 0, 0));           // a
  GP_TRY(gemm<R>(s, 0, 1, np1, nb, nb, 1.0, am, lb, S2, lb, 0.0, aV, lb, 0, 0, 0));           // a V_b
  GP_TRY(launch_symv_lower(Ki, lb, nb, mbp, bvec, s));
  GP_TRY(launch_dot(mbp, bvec, nb, c->scal + 6, s));
  // ---- moments / likelihood pieces with a = K_b K~_b^-1, per-point adjoints
  GP_TRY(launch_proj_moments(am, Kb, aV, lb, nb, mbp, c->Kvec, r, n1, A, lambda0, c->lam_m, c->lam_var, c->fvec, gm, gv,
                             c->upart, c->scal + 0, s));
  double* Ga = Kt;       // [np1][nb]
  double* GaKi = Kr;     // [np1][nb]
  GP_TRY(launch_proj_ga(Kb, aV, lb, nb, n1, np1, gm, gv, mbp, Ga, s));
  GP_TRY(gemm<R>(s, 0, 1, np1, nb, nb, 1.0, Ga, lb, Ki, lb, 0.0, GaKi, lb, 0, 0, 0));
  GP_TRY(gemm_splitk<R>(s, 1, 1, nb, nb, np1, 1.0, am, lb, GaKi, lb, c->Wbuf, lb, splitk_for(nb, nb, np1), c->TmpV + (int64_t)nb * nb,
                        (int64_t)c->np_cap * c->np_cap - (int64_t)nb * nb));   // P2 = a^T G_a K~_b^-1 (slabs behind P1 in TmpV)
  GP_TRY(launch_proj_gktb(Ki, S4, c->Wbuf, lb, nb, bvec, S3, s));                            // G_K~b
  GP_TRY(launch_proj_gkb(am, lb, nb, n1, np1, gv, GaKi, s));                                 // G_Kb (in place)
  // ---- the two adjoints:  W~ = sym(B G_K~b B^T) [np2 x np2],  W_K = G_Kb B^T [np1 x np2]
  GP_TRY(gemm<R>(s, 0, 1, np2, nb, nb, 1.0, Bp, lb, S3, lb, 0.0, Kt, lb, 0, 0, 0));           // B G_K~b  (G_a is dead)
  GP_TRY(gemm<R>(s, 0, 0, np2, np2, nb, 1.0, Kt, lb, Bp, lb, 0.0, c->Wbuf, l2, 0, 0, 0));
  GP_TRY(launch_symmetrize_avg(c->Wbuf, l2, np2, s));
  GP_TRY(gemm<R>(s, 0, 0, np1, np2, nb, 1.0, GaKi, lb, Bp, lb, 0.0, aV, l2, 0, 0, 0));        // W_K  (a V_b is dead)
  double* WK = aV;
  // ---- square pull-back on the inducing stimuli (no b b^T term, no dKvec term)
  GP_HIP(hipMemsetAsync(c->bv, 0, (size_t)c->np_cap * sizeof(R), s));
  GP_HIP(hipMemsetAsync(c->wl, 0, (size_t)c->np_cap * sizeof(R), s));
  GP_TRY(launch_adjoint(c->Wbuf, CosT, l2, c->bv, c->q2, n2, np2, Kb, c->upart, c->vpart, c->sumA_part, s));  // K_b is dead
  const int t64 = np2 / 64;
  GP_TRY(launch_adjoint_reduce(c->upart, c->vpart, c->sumA_part, t64, t64 * (t64 + 1) / 2, c->q2, c->wl, n2, np2, c->tvec,
                               c->rpad, c->scal + 7, s));
  GP_TRY(gemm<R>(s, 1, 1, np2, dp, np2, 1.0, Kb, l2, X2m, dp, 0.0, c->Ybuf, dp, 0, 0, 0));
  GP_TRY(launch_rowscale_add(c->Ybuf, dp, X2m, dp, c->tvec, np2, dp, s));
  auto xty = [&](const double* Xa, const double* Yb, int np, double* out) -> int {
    GemmArgsT<R> g{};
    g.A = Xa; g.B = Yb; g.C = c->Mpart; g.lda = dp; g.ldb = dp; g.ldc = dp;
    g.M = dp; g.N = dp; g.K = np; g.alpha = 1.0; g.beta = 0.0; g.a_kmajor = 1; g.b_kmajor = 1;
    g.batch = 1; g.split_k = c->split_k_M; g.sC = (int64_t)dp * dp;
    GP_TRY(launch_gemm(g, s));
    return launch_reduce_slices(c->Mpart, (int64_t)dp * dp, c->split_k_M, out, (int64_t)dp * dp, s);
  };
  GP_TRY(xty(X2m, c->Ybuf, np2, c->Mmat));
  // ---- rectangular pull-back (x, xtilde) with gvec = -g_v on the training side (dKvec term)
  GP_TRY(launch_scale_copy<R>(gvec, gv, n1, -1.0, s));
  double* t1 = c->tvec;
  double* t2 = c->tvec + c->np_cap;
  GP_TRY(launch_adjoint_rect(WK, l2, CosR, l2, c->q, c->q2, n1, n2, np1, np2, Kt, l2, c->upart, c->vpart, c->rect_part,
                             gvec, t1, t2, c->rpad, c->mpad, c->scal + 20, s));
  GP_TRY(gemm<R>(s, 0, 1, np1, dp, np2, 1.0, Kt, l2, X2m, dp, 0.0, c->Ybuf, dp, 0, 0, 0));
  GP_TRY(launch_rowscale_add(c->Ybuf, dp, X1m, dp, t1, np1, dp, s));
  GP_HIP(hipMemsetAsync(Zm, 0, (size_t)np2 * dp * sizeof(double), s));
  GP_TRY(launch_rowscale_add(Zm, dp, X2m, dp, t2, np2, dp, s));
  GP_TRY(xty(X1m, c->Ybuf, np1, c->dCpad));
  GP_TRY(launch_axpby_block<double>(c->Mmat, dp, c->dCpad, dp, dp, dp, 1.0, 1.0, s));
  GP_TRY(xty(X2m, Zm, np2, c->dCpad));
  GP_TRY(launch_axpby_block<double>(c->Mmat, dp, c->dCpad, dp, dp, dp, 1.0, 1.0, s));
  GP_TRY(launch_symmetrize_avg(c->Mmat, dp, dp, s));
  GP_TRY(launch_metric_contract(th, c->pix, d, n_rows, n_cols, c->Cmat, dp, c->Mmat, dp, c->scal + 10, c->upart, c->info + 3, s));
  GP_HIP(hipMemcpyAsync(c->scal_host, c->scal, 64 * sizeof(double), hipMemcpyDeviceToHost, s));
  GP_HIP(hipMemcpyAsync(c->info_host, c->info, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  GP_HIP(hipStreamSynchronize(s));
  const double* sc = c->scal_host;
  const double loglik = A * sc[0] + lambda0 * sc[1] - sc[2];
  const double KL = -0.5 * sc[40] + 0.5 * sc[3] + 0.5 * sc[6] + 0.5 * sc[5];
  const double sum_gvec = 0.5 * A * A * sc[2];                                               // -sum g_v
  out_host[0] = -(loglik - KL);
  out_host[1] = loglik;
  out_host[2] = KL;
  out_host[3] = th.sigma0 * (2.0 * sc[9] + 2.0 * sc[7]) + th.sigma0 * (2.0 * sc[20] + sc[21] + sc[22]) +
                2.0 * th.sigma0 * sum_gvec;
  out_host[4] = sc[13];
  out_host[5] = sc[14];
  out_host[6] = sc[11];
  out_host[7] = sc[12];
  out_host[8] = sc[10];
  out_host[9] = sc[3];
  out_host[10] = sc[40];
  out_host[11] = sc[5];
  out_host[12] = sc[6];
  out_host[13] = (double)d;
  out_host[14] = (double)c->info_host[0];
  out_host[15] = (double)c->info_host[1];
  if (c->info_host[0] != 0) {
    set_error("gpfit_fit_eval_sparse: Cholesky of the projected K_tilde failed (non-positive pivot)");
    return c->info_host[0];
  }
  if (c->info_host[1] != 0) {
    set_error("gpfit_fit_eval_sparse: Cholesky of V_b failed (non-positive pivot)");
    return c->info_host[1];
  }
  return 0;
}

// Wait for the evaluation enqueued on this context and assemble its 16 host scalars.
int fit_eval_finish(gpfit_ctx* c, double* out_host) {
  if (!c || !out_host || !c->pend.active) {
    set_error("gpfit_fit_eval_finish: nothing pending on this context");
    return -3;
  }
  c->pend.active = false;
  DeviceGuard device_guard(c->device);
  if (c->pend.use_done && c->pend.done) GP_HIP(hipEventSynchronize(c->pend.done));
  else GP_HIP(hipStreamSynchronize(c->pend.stream));
  const int n = c->pend.n, np = c->pend.np, want_grad = c->pend.want_grad;
  const double A = c->pend.A, lambda0 = c->pend.lambda0, sigma0 = c->pend.sigma0;

  const double* sc = c->scal_host;
  const double loglik = A * sc[0] + lambda0 * sc[1] - sc[2];                   // utils.py:1243
  // the identity padding of both factors contributes exactly (np - n) to ||L^-1 L_V||_F^2
  const double trKinvV = sc[5] - (double)(np - n);
  const double logdetV = sc[40];
  const double KL = -0.5 * logdetV + 0.5 * sc[3] + 0.5 * sc[6] + 0.5 * trKinvV;  // utils.py:1326
  out_host[0] = -(loglik - KL);                                                // utils.py:2087-2089
  out_host[1] = loglik;
  out_host[2] = KL;
  if (want_grad) {
    // d(loss)/d(theta) = dKL - dL (utils.py:2097-2099); metric rows come from the contraction,
    // the sigma_0 row from the closed form derived from utils.py:996-1004 / 1036.
    out_host[3] = sigma0 * (2.0 * sc[9] + 2.0 * sc[7]) - 2.0 * sigma0 * sc[8];
    out_host[4] = sc[13];  // eps_0x
    out_host[5] = sc[14];  // eps_0y
    out_host[6] = sc[11];  // -2log2beta
    out_host[7] = sc[12];  // -log2rho2
    out_host[8] = sc[10];  // Amp
  } else {
    for (int i = 0; i < 6; ++i) out_host[3 + i] = 0.0;
  }
  out_host[9] = sc[3];
  out_host[10] = logdetV;
  out_host[11] = trKinvV;
  out_host[12] = sc[6];
  out_host[13] = (double)c->pend.d;
  out_host[14] = (double)c->info_host[0];
  out_host[15] = (double)c->info_host[1];
  if (c->info_host[0] != 0) {
    set_error("Cholesky of K_tilde failed: non-positive pivot");
    return c->info_host[0];
  }
  if (c->info_host[1] != 0) {
    set_error("Cholesky of V failed: non-positive pivot");
    return c->info_host[1];
  }
  c->lv_valid = true;
  c->lv_n = n;
  c->lv_bytes = c->pend.elem_bytes;
  return 0;
}

}  // namespace gpfit

using namespace gpfit;

extern "C" {

int gpfit_fit_eval(gpfit_ctx* c, void* stream, const double* theta, const double* lower, const double* upper,
                   int n_rows, int n_cols, const double* X, int64_t ldx, int64_t N, const double* r,
                   const double* m, const double* V, int64_t ldv, double logA, double lambda0, int want_grad,
                   double* out_host, double* lam_m_out, double* lam_var_out, double* f_out) {
  return fit_eval_impl<double>(c, stream, theta, lower, upper, n_rows, n_cols, X, ldx, N, r, m, V, ldv, logA, lambda0,
                               want_grad, out_host, lam_m_out, lam_var_out, f_out);
}

int gpfit_fit_eval_f32(gpfit_ctx* c, void* stream, const double* theta, const double* lower, const double* upper,
                       int n_rows, int n_cols, const float* X, int64_t ldx, int64_t N, const float* r,
                       const float* m, const float* V, int64_t ldv, double logA, double lambda0, int want_grad,
                       double* out_host, float* lam_m_out, float* lam_var_out, float* f_out) {
  return fit_eval_impl<float>(c, stream, theta, lower, upper, n_rows, n_cols, X, ldx, N, r, m, V, ldv, logA, lambda0,
                              want_grad, out_host, lam_m_out, lam_var_out, f_out);
}

int gpfit_fit_eval_finish(gpfit_ctx* c, double* out_host) { return fit_eval_finish(c, out_host); }

int gpfit_fit_eval_batch(gpfit_ctx* const* ctxs, int n_units, void* stream, const double* theta, const double* lower,
                         const double* upper, int n_rows, int n_cols, const double* const* X, int64_t ldx, int64_t N,
                         const double* const* r, const double* const* m, const double* const* V, int64_t ldv,
                         const double* logA, const double* lambda0, int want_grad, double* out_host, int* rc_out) {
  return fit_eval_batch_impl<double>(ctxs, n_units, stream, theta, lower, upper, n_rows, n_cols, X, ldx, N, r, m, V, ldv,
                                     logA, lambda0, want_grad, out_host, rc_out);
}

int gpfit_fit_eval_batch_f32(gpfit_ctx* const* ctxs, int n_units, void* stream, const double* theta, const double* lower,
                             const double* upper, int n_rows, int n_cols, const float* const* X, int64_t ldx, int64_t N,
                             const float* const* r, const float* const* m, const float* const* V, int64_t ldv,
                             const double* logA, const double* lambda0, int want_grad, double* out_host, int* rc_out) {
  return fit_eval_batch_impl<float>(ctxs, n_units, stream, theta, lower, upper, n_rows, n_cols, X, ldx, N, r, m, V, ldv,
                                    logA, lambda0, want_grad, out_host, rc_out);
}

int gpfit_fit_eval_projected(gpfit_ctx* c, void* stream, const double* theta, const double* lower, const double* upper,
                             int n_rows, int n_cols, const double* X, int64_t ldx, int64_t N, const double* r,
                             const double* B, int64_t ldb, int64_t n_kept, const double* m_b, const double* V_b,
                             int64_t ldvb, double logA, double lambda0, double* out_host) {
  return fit_eval_projected_impl(c, stream, theta, lower, upper, n_rows, n_cols, X, ldx, N, r, B, ldb, n_kept, m_b, V_b,
                                 ldvb, logA, lambda0, out_host);
}

int gpfit_fit_eval_sparse(gpfit_ctx* c, void* stream, const double* theta, const double* lower, const double* upper,
                          int n_rows, int n_cols, const double* X, int64_t ldx, int64_t N, const double* Xtilde,
                          int64_t ldxt, int64_t Ntilde, const double* r, const double* B, int64_t ldb, int64_t n_kept,
                          const double* m_b, const double* V_b, int64_t ldvb, double logA, double lambda0,
                          double* out_host) {
  return fit_eval_sparse_impl(c, stream, theta, lower, upper, n_rows, n_cols, X, ldx, N, Xtilde, ldxt, Ntilde, r, B, ldb,
                              n_kept, m_b, V_b, ldvb, logA, lambda0, out_host);
}

int gpfit_grad_pullback(gpfit_ctx* c, void* stream, const double* theta, int n_rows, int n_cols, const double* X,
                        int64_t ldx, int64_t N, const double* W, int64_t ldw, const double* gvec, double* out6) {
  return grad_pullback_impl(c, stream, theta, n_rows, n_cols, X, ldx, N, W, ldw, gvec, out6);
}

int gpfit_ctx_create(int device, int64_t n_max, int64_t d_max, int64_t d_full_max, gpfit_ctx** out) {
  if (!out || n_max <= 0 || d_max <= 0) {
    set_error("gpfit_ctx_create: bad argument");
    return -3;
  }
  DeviceGuard device_guard(device);  // the caller's current device is restored on return
  {
    int cur = -1;
    GP_HIP(hipGetDevice(&cur));
    if (cur != device) {
      set_error("gpfit_ctx_create: cannot select the requested device");
      return -3;
    }
  }
  gpfit_ctx* c = new gpfit_ctx();
  c->device = device;
  c->np_cap = (int)round_up(n_max, TILE);
  c->dp_cap = (int)round_up(d_max, 32);
  c->dfull_cap = (int)std::max<int64_t>(d_full_max, d_max);
  const size_t np = c->np_cap, dp = c->dp_cap, nn = np * np;
  int rc = 0;
  auto A = [&](double** p, size_t cnt) { if (!rc) rc = dev_alloc(c, p, cnt); };
  A(&c->Kbuf, nn); A(&c->Cos, nn); A(&c->Lbuf, nn); A(&c->Libuf, nn); A(&c->Vbuf, nn); A(&c->LVbuf, nn);
  A(&c->LiVbuf, nn); A(&c->Tbuf, nn); A(&c->Zbuf, nn); A(&c->Wbuf, nn); A(&c->Abuf, nn); A(&c->Tmp, nn);
  A(&c->TmpV, nn);
  A(&c->Xt, dp * np); A(&c->Xm, np * dp); A(&c->XCt, dp * np); A(&c->Cmat, dp * dp); A(&c->Ybuf, np * dp);
  A(&c->Mpart, (size_t)c->split_k_M * dp * dp); A(&c->Mmat, dp * dp);
  A(&c->Xt2, dp * np); A(&c->XCt2, dp * np); A(&c->XDt, dp * np); A(&c->XDt2, dp * np); A(&c->dCpad, dp * dp);
  A(&c->q2, np); A(&c->dq1, np); A(&c->dq2, np); A(&c->hvec, np);
  A(&c->Kvec, np); A(&c->q, np); A(&c->lam_m, np); A(&c->lam_var, np); A(&c->fvec, np); A(&c->wl, np);
  A(&c->yv, np); A(&c->bv, np); A(&c->tvec, 2 * np); A(&c->mpad, np); A(&c->rpad, np);
  const size_t t64 = np / 64;
  A(&c->upart, t64 * np); A(&c->vpart, t64 * np); A(&c->sumA_part, t64 * (t64 + 1) / 2);
  A(&c->rect_part, t64 * t64);
  A(&c->frob_part, 33 * (np / TILE) * (np / TILE + 1) / 2); A(&c->trmv_part, (np / TRMV_ROWS + 1) * np);
  A(&c->scal, 64);
  for (int i = 0; i < 4 && !rc; ++i) {
    double* w = nullptr;
    rc = dev_alloc(c, &w, SK_WS_BYTES / sizeof(double));
    c->sk_ws[i] = w;
  }
  if (!rc) rc = dev_alloc(c, &c->pix, (size_t)c->dfull_cap);
  if (!rc) rc = dev_alloc(c, &c->info, 4);
  if (rc) {
    gpfit_ctx_destroy(c);
    return rc;
  }
  // pinned AND mapped into the device's address space: the group kernels read pix_host and write scal_host / info_host
  // directly (elementwise.hip: group_prepare_kernel, group_collect_kernel)
  constexpr unsigned HOST_FLAGS = hipHostMallocMapped | hipHostMallocPortable;
  GP_HIP(hipHostMalloc((void**)&c->scal_host, 64 * sizeof(double), HOST_FLAGS));
  GP_HIP(hipHostMalloc((void**)&c->pix_host, (size_t)c->dfull_cap * sizeof(int), HOST_FLAGS));
  GP_HIP(hipHostMalloc((void**)&c->info_host, 4 * sizeof(int), HOST_FLAGS));
  {
    // the V chain is the shorter of the two factorisations: give its stream the lowest priority so
    // that, whenever both have workgroups ready, the critical K~ chain is dispatched first
    int least = 0, greatest = 0;
    GP_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    const char* e = getenv("GPFIT_AUX_PRIO");  // tuning knob: 0 = default priority
    if (e && atoi(e) == 0) GP_HIP(hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
    else GP_HIP(hipStreamCreateWithPriority(&c->aux, hipStreamNonBlocking, least));
  }
  // (the side streams of the two chains are created on first use: ensure_side_streams)
  GP_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  GP_HIP(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  // the strict-upper tiles of every triangular work matrix are never written and must read as 0
  // wherever a dense GEMM touches them (Zbuf/Abuf are written in full; the others are only read
  // through triangular k ranges) -- zero everything once so no kernel ever sees garbage.
  for (double* p : {c->Kbuf, c->Cos, c->Lbuf, c->Libuf, c->Vbuf, c->LVbuf, c->LiVbuf, c->Tbuf, c->Zbuf, c->Wbuf,
                    c->Abuf, c->Tmp, c->TmpV})
    GP_HIP(hipMemset(p, 0, nn * sizeof(double)));
  GP_HIP(hipDeviceSynchronize());
  *out = c;
  return 0;
}

void gpfit_ctx_destroy(gpfit_ctx* c) {
  if (!c) return;
  DeviceGuard device_guard(c->device);
  (void)hipDeviceSynchronize();
  for (void* p : c->allocs) (void)hipFree(p);
  if (c->pend.done) (void)hipEventDestroy(c->pend.done);
  if (c->scal_host) (void)hipHostFree(c->scal_host);
  if (c->pix_host) (void)hipHostFree(c->pix_host);
  if (c->info_host) (void)hipHostFree(c->info_host);
  if (c->aux) (void)hipStreamDestroy(c->aux);
  for (int i = 0; i < 2; ++i) {
    if (c->side[i]) (void)hipStreamDestroy(c->side[i]);
    for (hipEvent_t e : c->side_ev[i]) (void)hipEventDestroy(e);
  }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  for (hipEvent_t e : c->phase_ev)
    if (e) (void)hipEventDestroy(e);
  delete c;
}

int gpfit_set_profile(gpfit_ctx* c, int on) {
  if (!c) return -3;
  c->profile = (on == 2) ? 2 : (on ? 1 : 0);
  return 0;
}

int gpfit_get_phases(gpfit_ctx* c, double* out8) {
  if (!c || !out8) return -3;
  if (!c->phase_valid) {
    set_error("gpfit_get_phases: no phase-timed evaluation (gpfit_set_profile(ctx, 2), then a synchronous gpfit_fit_eval with gradients)");
    return -3;
  }
  DeviceGuard device_guard(c->device);
  GP_HIP(hipDeviceSynchronize());
  for (int i = 0; i < 8; ++i) {
    float t = 0.f;
    GP_HIP(hipEventElapsedTime(&t, c->phase_ev[0], c->phase_ev[i]));
    out8[i] = t;
  }
  return 0;
}

double gpfit_last_enqueue_ms(gpfit_ctx* c) { return c ? c->last_enqueue_ms : -1.0; }

int gpfit_get_profile(gpfit_ctx* c, double* out16) {
  if (!c || !out16) return -3;
  for (int i = 0; i < 16; ++i) out16[i] = c->prof_out[i];
  return 0;
}

int gpfit_check_limits(const double* theta, const double* lower, const double* upper) {
  return check_limits(theta, lower, upper);
}

int gpfit_localker_mask(const double* theta, int n_rows, int n_cols, uint8_t* mask_host, int64_t* d_out) {
  if (!theta || n_rows <= 0 || n_cols <= 0) {
    set_error("gpfit_localker_mask: bad argument");
    return -3;
  }
  const int d = compute_mask(theta, n_rows, n_cols, mask_host, nullptr);
  if (d_out) *d_out = d;
  return 0;
}


}  // extern "C"
