// Workspace context of the GP fit library.  Internal header.
#pragma once
#include "kernels.h"
#include <vector>

struct gpfit_ctx {
  int device = 0;
  int np_cap = 0, dp_cap = 0, dfull_cap = 0;
  hipStream_t aux = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;

  // N x N work matrices (each np_cap^2 doubles)
  double *Kbuf = nullptr, *Cos = nullptr, *Lbuf = nullptr, *Libuf = nullptr, *Vbuf = nullptr, *LVbuf = nullptr,
         *LiVbuf = nullptr, *Tbuf = nullptr, *Zbuf = nullptr, *Wbuf = nullptr, *Abuf = nullptr, *Tmp = nullptr,
         *TmpV = nullptr;
  // N x d / d x d
  double *Xt = nullptr, *Xm = nullptr, *XCt = nullptr, *Cmat = nullptr, *Ybuf = nullptr, *Mpart = nullptr,
         *Mmat = nullptr, *Xt2 = nullptr, *XCt2 = nullptr, *XDt = nullptr, *XDt2 = nullptr, *dCpad = nullptr;
  // vectors (np_cap each unless noted)
  double *Kvec = nullptr, *q = nullptr, *lam_m = nullptr, *lam_var = nullptr, *fvec = nullptr, *wl = nullptr,
         *yv = nullptr, *bv = nullptr, *tvec = nullptr /* 2 np */, *mpad = nullptr, *rpad = nullptr,
         *q2 = nullptr, *dq1 = nullptr, *dq2 = nullptr, *hvec = nullptr;
  double *upart = nullptr, *vpart = nullptr, *sumA_part = nullptr, *frob_part = nullptr, *trmv_part = nullptr;
  double* rect_part = nullptr;  // (np/64)^2 per-tile sums of the rectangular adjoint
  void* sk_ws[4] = {nullptr, nullptr, nullptr, nullptr};  // stream-K partial-tile workspaces (main / aux / two side streams)
  // look-ahead: the first product of every inverse merge (tmp = L21 Li11) runs on a side stream of
  // its chain while the second half of the block is being factored (fit.hip:potrf_rec)
  hipStream_t side[2] = {nullptr, nullptr};
  std::vector<hipEvent_t> side_ev[2];
  int side_ev_next[2] = {0, 0};
  double* scal = nullptr;       // device scalars [64]
  double* scal_host = nullptr;  // pinned [64]
  int* pix = nullptr;           // device [dfull_cap]
  int* pix_host = nullptr;      // pinned
  int* info = nullptr;          // device [4]
  int* info_host = nullptr;     // pinned [4]
  std::vector<void*> allocs;
  int split_k_M = 32;

  // optional per-launch event timing of the dominant kernels (bench.py roofline leg)
  int profile = 0;
  struct ProfRec { hipEvent_t a, b; double flops; int kind; };  // kind 0 gemm T=128, 1 leaf, 2 gram, 3 gemm T<128
  std::vector<ProfRec> prof;
  std::vector<hipEvent_t> ev_pool;
  double prof_out[16] = {0};
  // profile == 2: phase timing only (eight events per fit, none inside the factorisations): 0 start,
  // 1 kernel build + moments done (fork), 2 K~ chain done, 3 V chain done (aux stream), 4 T and its
  // norm done, 5 Q = I - T T^T done, 6 two-sided product done, 7 end
  hipEvent_t phase_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool phase_valid = false;
  double last_enqueue_ms = 0.0;  // host time spent enqueuing the last fit_eval
  // evaluation enqueued but not yet collected (gpfit_fit_eval with the async flag / _finish)
  // done: recorded behind the result copies of a group (gpfit_fit_eval_batch), so that collecting a unit waits for
  // ITS group only and a second group can already run on the same stream (the single-unit call syncs the stream)
  struct Pending { bool active = false; hipStream_t stream = nullptr; double A = 0, lambda0 = 0, sigma0 = 0;
                   int n = 0, np = 0, d = 0, want_grad = 0, elem_bytes = 8; hipEvent_t done = nullptr;
                   bool use_done = false; } pend;

  // cached state of the last upload / evaluation (used by estep / predict entry points)
  int cur_n = 0, cur_np = 0, cur_d = 0, cur_dp = 0;
  bool lv_valid = false;  // LVbuf / scal[40] hold the factor and log-det of the last V
  bool lv32_valid = false;  // Vbuf holds the single-precision copy of that factor (mixed-precision mode)
  int lv_n = 0;
  int lv_bytes = 0;       // element size the cached factor was computed in
};

namespace gpfit {

// Every context entry point runs on the context's device whatever the caller's current device is,
// and restores the caller's device on return.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = (hipSetDevice(device) == hipSuccess);
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};
// First statements of a context entry point (after its argument check): refuse to touch the
// workspace of an evaluation that is still in flight, then pin the device.
#define GP_CTX_ENTER(c, name)                                                                        \
  if ((c)->pend.active) {                                                                            \
    gpfit::set_error(name ": an asynchronous evaluation is pending on this context (collect it with " \
                          "gpfit_fit_eval_finish first)");                                           \
    return -3;                                                                                       \
  }                                                                                                  \
  gpfit::DeviceGuard _device_guard((c)->device)

// profiling scope: when a context with profile=1 is evaluating, launches are bracketed by events
void prof_begin(gpfit_ctx* c);
void prof_end(gpfit_ctx* c);   // synchronises and fills c->prof_out
struct ProfScope {
  hipStream_t s; double flops; int kind; hipEvent_t a = nullptr;
  ProfScope(hipStream_t s, double flops, int kind);
  ~ProfScope();
};
template <typename R> double gemm_flops(const GemmArgsT<R>& g);

template <typename R>
struct CholBufsT {
  R* A;    // input, lower triangle; destroyed
  R* L;    // output factor
  R* Li;   // output inverse blocks (full inverse when need_inv at the top)
  R* Tmp;  // scratch, same shape
  int64_t ld;
  int* info;
  int ws = 0;   // stream-K workspace id (1 for the factorisation running on the aux stream)
  void* sk_ws = nullptr;  // that workspace (owned by the context)
  // optional look-ahead resources of this chain (nullptr / 0: everything on the one stream)
  gpfit_ctx* ctx = nullptr;
  int chain = 0;           // index into ctx->side / side_ev
  int side_min = 0;        // blocks of at least this size put their merge product on the side stream
  int half_occ = 0;        // bit 0: this chain's own 128-tile launches run one workgroup per CU; bit 1: its side-stream products do
  // optional: record mark_ev on the chain's stream once the leading mark_n x mark_n block is factored
  // (the other chain can be started there, so that its latency-bound leaf stretches meet this chain's
  // large products instead of this chain's leaf stretches)
  hipEvent_t mark_ev = nullptr;
  int mark_n = 0;
};
using CholBufs = CholBufsT<double>;
// Recursive blocked Cholesky of the n x n diagonal block at offset r0 (n a multiple of 128),
// built entirely from the MFMA GEMM and the 128 x 128 leaf.  need_inv = 1: the full inverse of
// the factor is assembled on the way (L^-1 costs n^3/3 more; 0: only n^3/12 for the sub-block
// inverses the solves need); 2: the inverses of the two diagonal half blocks but not the
// off-diagonal block [L^-1]21 (for callers that apply L^-1 block-wise, n^3/8 less).
template <typename R>
int potrf_rec(const CholBufsT<R>& B, int r0, int n, int need_inv, hipStream_t s);

// Several chains of the same size factored in lock step (fit.hip:potrf_lockstep): chain b factors A[b] (lower
// triangle, destroyed) into L[b] with the inverse blocks in Li[b], scratch Tmp[b], LAPACK info in info[b].
template <typename R>
struct CholBatchT {
  int nb = 0;
  R* A[GEMM_MAXB];
  R* L[GEMM_MAXB];
  R* Li[GEMM_MAXB];
  R* Tmp[GEMM_MAXB];
  int* info[GEMM_MAXB];
  int64_t ld = 0;
  int ws = 0;
  void* sk_ws = nullptr;     // stream-K workspace of the launches issued chain by chain (one stream: shared)
  gpfit_ctx* ctx = nullptr;  // look-ahead resources (side stream `chain` of this context), or nullptr
  int chain = 0;
  int side_min = 0;
};
// need: bit b = chain b needs the full inverse of its block (need_inv of potrf_rec); the diagonal sub-block
// inverses every chain's solves need are always formed.
template <typename R>
int potrf_lockstep(const CholBatchT<R>& B, int r0, int n, uint32_t need, hipStream_t s);

}  // namespace gpfit
