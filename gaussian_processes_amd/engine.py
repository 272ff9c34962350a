"""Thin Python host layer over the C ABI: one ``GPFitEngine`` = one ``gpfit_ctx`` on one GPU.

torch is used for device memory and the current HIP stream only; every number is
produced by the hand-written HIP kernels behind ``libgpfit_mi355x.so``."""
from __future__ import annotations

import ctypes
import math

import torch

from . import _lib
from .synthetic import THETA_KEYS


def _grid(n_px_side):
    if isinstance(n_px_side, (tuple, list)):
        return int(n_px_side[0]), int(n_px_side[1])
    n = int(round(float(n_px_side)))
    return n, n


def _scalar(v) -> float:
    return float(v.item()) if hasattr(v, "item") else float(v)


def theta_vec(theta):
    if isinstance(theta, dict):
        return [_scalar(theta[k]) for k in THETA_KEYS]
    return [_scalar(v) for v in theta]


class GPFitEngine:
    """Workspace + entry points for the GP fit hot path on one MI355X."""

    def __init__(self, n_max: int, d_max: int, d_full_max: int | None = None, device: int = 0):
        if not torch.cuda.is_available():
            raise _lib.GpfitError("GPFitEngine needs a GPU: the GP fit path has no CPU fallback")
        self.lib = _lib.load()
        self.device = int(device)
        self.tdev = torch.device("cuda", self.device)
        self.n_max, self.d_max = int(n_max), int(d_max)
        self.d_full_max = int(d_full_max or d_max)
        ctx = ctypes.c_void_p()
        _lib.check(self.lib.gpfit_ctx_create(self.device, self.n_max, self.d_max, self.d_full_max,
                                            ctypes.byref(ctx)), "gpfit_ctx_create")
        self._ctx = ctx

    def close(self):
        if getattr(self, "_ctx", None):
            self.lib.gpfit_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.tdev).cuda_stream)

    def _dev(self, t, name, dtype=torch.float64):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype):
            raise TypeError(f"{name} must be a {dtype} CUDA tensor")
        if t.dim() == 2 and t.stride(1) != 1:
            t = t.contiguous()
        if t.dim() == 1 and t.stride(0) != 1:
            t = t.contiguous()
        return t

    def set_profile(self, on):
        """0 / False off, 1 / True per-launch events of the dominant kernels, 2 phase timing only."""
        _lib.check(self.lib.gpfit_set_profile(self._ctx, int(on)), "gpfit_set_profile")

    def get_phases(self):
        """Milliseconds from the start of the last phase-timed evaluation to each phase boundary."""
        out = (ctypes.c_double * 8)()
        _lib.check(self.lib.gpfit_get_phases(self._ctx, out), "gpfit_get_phases")
        names = ("start", "kernel_build_done", "chol_K_done", "chol_V_done", "T_done", "Q_done", "W_done", "end")
        return {k: round(out[i], 4) for i, k in enumerate(names)}

    def get_profile(self):
        out = (ctypes.c_double * 16)()
        _lib.check(self.lib.gpfit_get_profile(self._ctx, out), "gpfit_get_profile")
        return {"gemm_ms": out[0], "gemm_flops": out[1], "gemm_launches": int(out[2]), "leaf_ms": out[3],
                "leaf_launches": int(out[4]), "gram_ms": out[5], "gram_flops": out[6],
                "small_gemm_ms": out[8], "small_gemm_flops": out[9], "small_gemm_launches": int(out[10]),
                "largest_gemm_ms": out[11], "largest_gemm_flops": out[12]}

    def last_enqueue_ms(self) -> float:
        return float(self.lib.gpfit_last_enqueue_ms(self._ctx))

    def mask(self, theta, n_px_side):
        """Pixel mask of localker (utils.py:880-883) as a host bool tensor, and its count."""
        rows, cols = _grid(n_px_side)
        buf = (ctypes.c_uint8 * (rows * cols))()
        d = ctypes.c_int64()
        _lib.check(self.lib.gpfit_localker_mask(_lib.darr(theta_vec(theta)), rows, cols, buf, ctypes.byref(d)),
                   "gpfit_localker_mask")
        return torch.tensor(list(buf), dtype=torch.bool), int(d.value)

    def fit_eval(self, theta, lower, upper, n_px_side, X, r, m, V, logA, lambda0, want_grad=True,
                 want_vectors=True, reuse_V=False, grad_precision="native"):
        """One evaluation of the M-step closure (utils.py:2017-2112), full-rank regime.

        Returns a dict with ``loss`` (= -logmarginal), ``loglik``, ``KL``, ``grad`` (dict in the
        reference's key order, d loss / d theta), diagnostics, and the device vectors
        ``lam_m, lam_var, f``.  float32 inputs select the fp32 instance of the library
        (hyperparameter-grid configuration); float64 is the reference's precision.  Out-of-box theta returns loss = inf and grad = inf (reference
        behaviour); a failed Cholesky raises ``GpfitError``.  ``reuse_V=True`` promises that V is
        the matrix of the previous call on this engine (constant during an M-step) so that its
        factorisation is not repeated.  ``grad_precision="f32"`` (float64 inputs) keeps the kernel build, both
        factorisations, the log-determinants and the likelihood in fp64 and runs the N^3 products T, Q, W and the
        pull-back in fp32 -- the trace term of the KL comes from the fp32 T (2e-9 on the loss at N = 8192)."""
        return self.fit_eval_finish(self.fit_eval_async(theta, lower, upper, n_px_side, X, r, m, V, logA, lambda0,
                                                        want_grad, want_vectors, reuse_V, _sync=True,
                                                        grad_precision=grad_precision))

    def fit_eval_async(self, theta, lower, upper, n_px_side, X, r, m, V, logA, lambda0, want_grad=True,
                       want_vectors=True, reuse_V=False, _sync=False, grad_precision="native"):
        """Enqueue one evaluation on the current stream without waiting for it; returns a ticket
        for :meth:`fit_eval_finish`.  One evaluation may be pending per engine: independent units
        (cells, theta points) are overlapped by alternating between engines on different streams
        (``multi.evaluate_units``)."""
        rows, cols = _grid(n_px_side)
        dtype = X.dtype if isinstance(X, torch.Tensor) and X.dtype == torch.float32 else torch.float64
        X, r, m, V = (self._dev(X, "X", dtype), self._dev(r, "r", dtype), self._dev(m, "m", dtype),
                      self._dev(V, "V", dtype))
        N = X.shape[0]
        if X.shape[1] != rows * cols:
            raise ValueError(f"X has {X.shape[1]} pixels but the grid is {rows}x{cols}")
        out = (ctypes.c_double * 16)()
        lam_m = lam_var = f = None
        ptrs = [None, None, None]
        if want_vectors:
            lam_m = torch.empty(N, dtype=dtype, device=self.tdev)
            lam_var = torch.empty_like(lam_m)
            f = torch.empty_like(lam_m)
            ptrs = [lam_m.data_ptr(), lam_var.data_ptr(), f.data_ptr()]
        lo = _lib.darr(theta_vec(lower)) if lower is not None else None
        up = _lib.darr(theta_vec(upper)) if upper is not None else None
        entry = self.lib.gpfit_fit_eval_f32 if dtype == torch.float32 else self.lib.gpfit_fit_eval
        if grad_precision not in ("native", "f32"):
            raise ValueError("grad_precision must be 'native' or 'f32'")
        if grad_precision == "f32" and dtype != torch.float64:
            raise ValueError("grad_precision='f32' is the mixed mode of the fp64 entry point: pass float64 inputs")
        flags = (1 if want_grad else 0) | (2 if reuse_V else 0) | (0 if _sync else 4) | (8 if grad_precision == "f32" else 0)
        rc = entry(self._ctx, self._stream(), _lib.darr(theta_vec(theta)), lo, up, rows, cols,
                   X.data_ptr(), X.stride(0), N, r.data_ptr(), m.data_ptr(), V.data_ptr(),
                   V.stride(0), float(logA), float(lambda0), flags, out, ptrs[0], ptrs[1], ptrs[2])
        pending = (rc == 0) and not _sync
        if rc < 0 and rc != -2:
            _lib.check(rc, "gpfit_fit_eval")
        # the ticket keeps the operand tensors alive until the evaluation has been collected
        return {"rc": rc, "out": out, "pending": pending, "keep": (X, r, m, V), "lam_m": lam_m, "lam_var": lam_var, "f": f}

    def fit_eval_finish(self, ticket):
        """Wait for the evaluation behind ``ticket`` and return the result dict of :meth:`fit_eval`."""
        rc, out = ticket["rc"], ticket["out"]
        if ticket["pending"]:
            rc = self.lib.gpfit_fit_eval_finish(self._ctx, out)
            ticket["pending"] = False
        if rc > 0:
            raise _lib.GpfitError(f"gpfit_fit_eval: {_lib.last_error()} (info={rc})")
        _lib.check(rc, "gpfit_fit_eval")
        return {
            "loss": out[0], "loglik": out[1], "KL": out[2],
            "grad": {k: out[3 + i] for i, k in enumerate(THETA_KEYS)},
            "logdet_K": out[9], "logdet_V": out[10], "tr_KinvV": out[11], "mKinvm": out[12],
            "d": int(out[13]) if rc == 0 else 0, "in_bounds": rc == 0,
            "lam_m": ticket["lam_m"], "lam_var": ticket["lam_var"], "f": ticket["f"],
        }


MAX_GROUP = 16  # units per gpfit_fit_eval_batch call (2 chains each; GEMM_MAXB = 32 problems per pointer batch)


def fit_eval_group_begin(engines, thetas, lower, upper, n_px_side, X, r, m, V, logA, lambda0, want_grad=True, reuse_V=False,
                         grad_precision="native"):
    """``len(thetas)`` independent units of the same N in ONE call (``gpfit_fit_eval_batch``): unit u on
    ``engines[u]`` with ``thetas[u]``, ``r[u]``, ``m[u]``, ``V[u]`` (``X``, ``r``, ``m``, ``V``, ``logA``, ``lambda0`` may
    each be one object shared by every unit or a list with one entry per unit).  The Cholesky recursions of all
    units run in lock step (shared launches on the latency-bound levels); results are bit-identical to
    ``engines[u].fit_eval`` unit by unit.  Enqueues only and returns a handle for :func:`fit_eval_group_finish`, which
    returns the list of result dicts of :meth:`GPFitEngine.fit_eval` (without the per-point vectors); with two sets of
    engines the next group is enqueued before the previous one is collected and the GPU never waits for the host
    (``multi.evaluate_units_grouped``)."""
    nu = len(thetas)
    if not 1 <= nu <= MAX_GROUP or len(engines) < nu:
        raise ValueError(f"fit_eval_group: 1 .. {MAX_GROUP} units per call, one engine per unit")
    e0 = engines[0]
    rows, cols = _grid(n_px_side)

    def per_unit(x):
        return list(x) if isinstance(x, (list, tuple)) else [x] * nu

    Xs, rs, ms, Vs = per_unit(X), per_unit(r), per_unit(m), per_unit(V)
    dtype = Xs[0].dtype if Xs[0].dtype == torch.float32 else torch.float64
    Xs = [e0._dev(t, "X", dtype) for t in Xs]
    rs = [e0._dev(t, "r", dtype) for t in rs]
    ms = [e0._dev(t, "m", dtype) for t in ms]
    Vs = [e0._dev(t, "V", dtype) for t in Vs]
    N = Xs[0].shape[0]
    if any(t.shape != Xs[0].shape or t.stride(0) != Xs[0].stride(0) for t in Xs) or Xs[0].shape[1] != rows * cols:
        raise ValueError("fit_eval_group: every unit needs stimuli of the same shape, matching the pixel grid")
    if any(t.shape != (N, N) or t.stride(0) != Vs[0].stride(0) for t in Vs):
        raise ValueError("fit_eval_group: every V must be N x N with the same row stride")
    if grad_precision not in ("native", "f32") or (grad_precision == "f32" and dtype != torch.float64):
        raise ValueError("grad_precision must be 'native' or, for float64 inputs, 'f32'")
    logAs = [float(_scalar(v)) for v in per_unit(logA)]
    lam0s = [float(_scalar(v)) for v in per_unit(lambda0)]
    ctxs = (ctypes.c_void_p * nu)(*[e._ctx for e in engines[:nu]])

    def ptrs(ts):
        return (ctypes.c_void_p * nu)(*[t.data_ptr() for t in ts])

    theta = _lib.darr([v for th in thetas for v in theta_vec(th)])
    lo = _lib.darr(theta_vec(lower)) if lower is not None else None
    up = _lib.darr(theta_vec(upper)) if upper is not None else None
    out = (ctypes.c_double * (16 * nu))()
    rcs = (ctypes.c_int * nu)()
    flags = (1 if want_grad else 0) | (2 if reuse_V else 0) | (8 if grad_precision == "f32" else 0)
    entry = e0.lib.gpfit_fit_eval_batch_f32 if dtype == torch.float32 else e0.lib.gpfit_fit_eval_batch
    rc = entry(ctxs, nu, e0._stream(), theta, lo, up, rows, cols, ptrs(Xs), Xs[0].stride(0), N, ptrs(rs), ptrs(ms), ptrs(Vs),
               Vs[0].stride(0), _lib.darr(logAs), _lib.darr(lam0s), flags, out, rcs)
    _lib.check(rc, "gpfit_fit_eval_batch")
    tickets = []
    for u in range(nu):
        o = (ctypes.c_double * 16)(*out[16 * u:16 * u + 16])
        tickets.append({"rc": int(rcs[u]), "out": o, "pending": rcs[u] == 0, "keep": (Xs[u], rs[u], ms[u], Vs[u]),
                        "lam_m": None, "lam_var": None, "f": None})
    return {"engines": list(engines[:nu]), "tickets": tickets}


def fit_eval_group_finish(handle):
    """Collect the group enqueued by :func:`fit_eval_group_begin`: waits for THAT group only (a completion event
    recorded behind its results), so a second group -- on other engines -- may already be running on the stream."""
    results, first_error = [], None
    for eng, ticket in zip(handle["engines"], handle["tickets"]):   # every pending unit is collected, also behind a failed one
        try:
            results.append(eng.fit_eval_finish(ticket))
        except _lib.GpfitError as err:
            results.append(None)
            first_error = first_error or err
    if first_error is not None:
        raise first_error
    return results


def fit_eval_group(engines, thetas, lower, upper, n_px_side, X, r, m, V, logA, lambda0, want_grad=True, reuse_V=False,
                   grad_precision="native"):
    """:func:`fit_eval_group_begin` and :func:`fit_eval_group_finish` in one call."""
    return fit_eval_group_finish(fit_eval_group_begin(engines, thetas, lower, upper, n_px_side, X, r, m, V, logA, lambda0,
                                                      want_grad=want_grad, reuse_V=reuse_V, grad_precision=grad_precision))


def fits_flops(N: int, d: int) -> float:
    """Algorithmic flops of one unit of work, SURVEY.md 8(d): (14/3)N^3 + 4N^2 d + 4 N d^2."""
    return (14.0 / 3.0) * N ** 3 + 4.0 * N * N * d + 4.0 * N * d * d
