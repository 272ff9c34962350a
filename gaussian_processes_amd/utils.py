"""Drop-in host module for the GP fit path of ``Spatial_GP_repo/utils.py``.

Same function names, argument meaning and error behaviour as the reference for the path
SURVEY.md section 8 scopes (``localker``, ``acosker``, ``lambda_moments``, ``compute_KL_div``,
``Estep``, ``varGP``, ``test`` ...), but every kernel / factorisation / solve runs in the
hand-written HIP library ``libgpfit_mi355x.so`` through ctypes.  torch supplies device
memory, streams, the L-BFGS drivers and, for the eigen-stabilisation of matrices below 256 rows or whenever the
eigh-free routes decline (an eigenvalue on the threshold), ``torch.linalg.eigh``.

There is no CPU fallback: without a GPU or without the built library every entry point raises.
Out of scope (SURVEY.md section 2 rows 10-14,16-19): plotting, persistence (``save_model`` / ``load_model``: the
reference's own functions pickle the ``fit_model`` dict this module returns), the dataset container, and the
reference's dead code (``utility`` / ``get_utility`` scalar variants, ``block_matrix_inverse``, ``updateA``,
``linker``: no caller in the reference's own files).
"""
from __future__ import annotations

import copy
import ctypes
import math
import threading
import warnings

import torch

from . import _lib
from .engine import GPFitEngine, _grid, theta_vec
from . import eigtop
from .synthetic import THETA_KEYS

torch.set_grad_enabled(False)  # reference utils.py:2 (analytic gradients only)

TORCH_DTYPE = torch.float64          # utils.py:31
torch.set_default_dtype(TORCH_DTYPE)  # utils.py:33 -- the reference makes float64 the global default on
#                                       import and its notebooks rely on it (torch.tensor(0.05) is fp64)
MIN_TOLERANCE = 1.e-11               # utils.py:37
EIGVAL_TOL = 1.e-4                   # utils.py:39 (module global read at call time, as in the reference)
PI32 = 3.1415927410125732            # utils.py:25: float32-rounded pi

# dC dict key order of the reference (utils.py:910) == matrix order of gpfit_localker
DC_KEYS = ("Amp", "-2log2beta", "-log2rho2", "eps_0x", "eps_0y")


_HAVE_GPU = [False]


def _device():
    if not _HAVE_GPU[0]:       # (asked once: torch.cuda.is_available() costs 2 us and this is called 30 000 times per fit)
        if not torch.cuda.is_available():
            raise _lib.GpfitError("gaussian_processes_amd.utils needs an MI355X: there is no CPU fallback")
        _HAVE_GPU[0] = True
    return torch.device("cuda", torch.cuda.current_device())


class _EnginePool:
    """One workspace per (device, host thread): the C context is not re-entrant, and the
    reference's active-learning notebook scores candidates on a second ``threading.Thread`` while
    the main thread fits (one_cell_active_training.ipynb:2446-2461).

    A workspace only ever grows.  Growing allocates a new context and drops the pool's reference
    to the old one, which is destroyed when its last holder lets go -- an engine a caller still
    holds stays valid.  ``varGP`` / ``test`` size the pool once up front (``reserve``), so no
    reallocation happens in the middle of a fit."""

    def __init__(self):
        self.eng = {}
        self.lock = threading.Lock()

    def get(self, n, d, d_full=None):
        dev = _device()
        key = (dev.index, threading.get_ident())
        d_full = int(d_full or d)
        with self.lock:
            e = self.eng.get(key)
            if e is None or e.n_max < n or e.d_max < d or e.d_full_max < d_full:
                n_cap = max(n, e.n_max if e else 0)
                d_cap = max(d, e.d_max if e else 0)
                f_cap = max(d_full, e.d_full_max if e else 0)
                if e is None:
                    alive = {t.ident for t in threading.enumerate()}
                    for k in [k for k in self.eng if k[1] not in alive]:
                        del self.eng[k]          # contexts of threads that have ended
                self.eng[key] = None             # release the old workspace before allocating the new one
                del e                            # (unless a caller still holds it)
                e = GPFitEngine(n_cap, d_cap, f_cap, device=dev.index)
                self.eng[key] = e
            return e


_POOL = _EnginePool()


def get_engine(n, d, d_full=None) -> GPFitEngine:
    return _POOL.get(int(n), int(d), d_full)


def reserve(n, d, d_full=None) -> GPFitEngine:
    """Size this thread's workspace for problems up to ``n`` stimuli and ``d`` (masked) / ``d_full``
    (image) pixels in one allocation."""
    return _POOL.get(int(n), int(d), d_full)


def _scalar(v) -> float:
    return float(v.item()) if hasattr(v, "item") else float(v)


def _cu(t, name="tensor"):
    """float64 contiguous CUDA view of ``t`` (moved if it lives on the host)."""
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t, dtype=TORCH_DTYPE)
    dev = _device()
    if t.dtype != TORCH_DTYPE or t.device != dev:
        t = t.to(device=dev, dtype=TORCH_DTYPE)
    return t.contiguous()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ------------------------------------------------------------------ kernel functions
def localker(theta, theta_higher_lims, theta_lower_lims, n_px_side, grad=False):
    """Spatial metric ``C`` (reference utils.py:861-914).  Returns ``(C, mask)`` or
    ``(C, mask, dC)`` with dC a dict keyed Amp, -2log2beta, -log2rho2, eps_0x, eps_0y.
    Raises ValueError when a hyperparameter is outside its limits (utils.py:865-867).
    ``n_px_side`` may also be ``(n_rows, n_cols)`` (rectangular generalisation)."""
    lib = _lib.load()
    th = _lib.darr(theta_vec(theta))
    lo = _lib.darr([_scalar(theta_lower_lims[k]) for k in THETA_KEYS])
    up = _lib.darr([_scalar(theta_higher_lims[k]) for k in THETA_KEYS])
    if lib.gpfit_check_limits(th, lo, up) != 0:
        raise ValueError(_lib.last_error())
    rows, cols = _grid(n_px_side)
    mbuf = (ctypes.c_uint8 * (rows * cols))()
    dcount = ctypes.c_int64()
    _lib.check(lib.gpfit_localker_mask(th, rows, cols, mbuf, ctypes.byref(dcount)), "gpfit_localker_mask")
    d = int(dcount.value)
    dev = _device()
    mask = torch.frombuffer(bytearray(mbuf), dtype=torch.uint8).to(torch.bool).to(dev)
    if d == 0:
        raise ValueError("localker: the pixel mask is empty")
    eng = get_engine(1, d, rows * cols)
    C = torch.empty((d, d), dtype=TORCH_DTYPE, device=dev)
    dCbuf = torch.empty((5, d, d), dtype=TORCH_DTYPE, device=dev) if grad else None
    _lib.check(lib.gpfit_localker(eng._ctx, _stream(), th, rows, cols, mbuf, d, C.data_ptr(),
                                  dCbuf.data_ptr() if grad else None), "gpfit_localker")
    if not grad:
        return C, mask
    return C, mask, {k: dCbuf[i] for i, k in enumerate(DC_KEYS)}


def _pack_dC(dC, d, dev):
    """dict of d x d matrices -> contiguous [5][d][d] in the library's order (or None)."""
    if dC is None:
        return None
    if all(k in dC for k in DC_KEYS):
        first = dC[DC_KEYS[0]]
        base = getattr(first, "_base", None)
        if (isinstance(base, torch.Tensor) and base.dim() == 3 and base.shape[0] == 5 and base.is_contiguous()
                and all(dC[k]._base is base and dC[k].data_ptr() == base[i].data_ptr()
                        for i, k in enumerate(DC_KEYS))):
            return base  # exactly what localker returned: no copy
        return torch.stack([_cu(dC[k]) for k in DC_KEYS]).contiguous()
    raise KeyError(f"dC must hold the keys {DC_KEYS}")


def acosker(theta, x1, x2=None, C=None, dC=None, diag=False):
    """Arc-cosine kernel (reference utils.py:939-1050).

    ``x1[n1, nx]``, ``x2[n2, nx]`` already masked; ``C[nx, nx]``.  diag=False returns
    ``K[n1, n2]`` (symmetrised when n1 == n2) and, if dC is given, ``(K, dK)`` with dK a dict
    over the six hyperparameters.  diag=True returns the vector ``x_i C x_i + sigma_0^2``."""
    lib = _lib.load()
    dev = _device()
    s0 = _scalar(theta["sigma_0"])
    x1 = _cu(x1)
    if x1.dim() == 1:
        x1 = x1[None, :]
    n1, d = x1.shape
    if C is None:
        C = torch.eye(n1, dtype=TORCH_DTYPE, device=dev)  # utils.py:973 (only meaningful when n1 == nx)
    C = _cu(C)
    if C.shape != (d, d):
        raise RuntimeError(f"acosker: C is {tuple(C.shape)} but the inputs have {d} pixels")
    dCbuf = _pack_dC(dC, d, dev)
    if diag:
        eng = get_engine(n1, d)
        Kvec = torch.empty(n1, dtype=TORCH_DTYPE, device=dev)
        dKv = torch.empty((6, n1), dtype=TORCH_DTYPE, device=dev) if dC is not None else None
        _lib.check(lib.gpfit_acosker_diag(eng._ctx, _stream(), s0, x1.data_ptr(), x1.stride(0), n1, d,
                                          C.data_ptr(), C.stride(0), dCbuf.data_ptr() if dCbuf is not None else None,
                                          Kvec.data_ptr(), dKv.data_ptr() if dKv is not None else None),
                   "gpfit_acosker_diag")
        if dC is None:
            return Kvec if n1 > 1 else Kvec.squeeze()
        return Kvec, {k: dKv[i] for i, k in enumerate(THETA_KEYS)}
    same = x2 is None or x2 is x1
    x2 = x1 if same else _cu(x2)
    if x2.dim() == 1:
        x2 = x2[None, :]
    if not same and x2.data_ptr() == x1.data_ptr() and x2.shape == x1.shape:
        x2 = x1
    n2 = x2.shape[0]
    if x2.shape[1] != d:
        raise RuntimeError("acosker: x1 and x2 have different numbers of pixels")
    eng = get_engine(max(n1, n2), d)
    K = torch.empty((n1, n2), dtype=TORCH_DTYPE, device=dev)
    dK = torch.empty((6, n1, n2), dtype=TORCH_DTYPE, device=dev) if dC is not None else None
    _lib.check(lib.gpfit_acosker(eng._ctx, _stream(), s0, x1.data_ptr(), x1.stride(0), n1, x2.data_ptr(),
                                 x2.stride(0), n2, d, C.data_ptr(), C.stride(0),
                                 dCbuf.data_ptr() if dCbuf is not None else None, K.data_ptr(), K.stride(0),
                                 dK.data_ptr() if dK is not None else None), "gpfit_acosker")
    if dC is None:
        return K
    return K, {k: dK[i] for i, k in enumerate(THETA_KEYS)}


# ------------------------------------------------------------------ dense helpers on the MFMA GEMM
def _pad2(t, rows, cols):
    if t.shape[0] == rows and t.shape[1] == cols and t.is_contiguous():
        return t
    out = torch.zeros((rows, cols), dtype=TORCH_DTYPE, device=t.device)
    out[: t.shape[0], : t.shape[1]] = t
    return out


def _ceil(x, m):
    return (x + m - 1) // m * m


def matmul(A, B, transA=False, transB=False, alpha=1.0):
    """``alpha * op(A) @ op(B)`` on the fp64 MFMA GEMM (``gpfit_dgemm``): the torch ``@`` call
    sites of the reference's path.  1-D operands are treated as column/row vectors."""
    lib = _lib.load()
    # fast path (the subspace solver and the E-steps issue thousands of these per fit: the wrapper's own checks were a
    # quarter of their cost): 2-D float64 device operands, contiguous, already aligned -- straight to the C ABI
    if (type(A) is torch.Tensor and type(B) is torch.Tensor and A.dim() == 2 and B.dim() == 2 and A.is_cuda and B.is_cuda
            and A.dtype is TORCH_DTYPE and B.dtype is TORCH_DTYPE and A.is_contiguous() and B.is_contiguous()):
        M, K = (A.shape[1], A.shape[0]) if transA else A.shape
        K2, N = (B.shape[1], B.shape[0]) if transB else B.shape
        if K == K2 and not (K & 15) and not (M & 1) and not (N & 1) and A.device == B.device:
            C = torch.empty((M, N), dtype=TORCH_DTYPE, device=A.device)
            _lib.check(lib.gpfit_dgemm(_stream(), 1 if transA else 0, 0 if transB else 1, M, N, K, float(alpha),
                                       A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), 0.0, C.data_ptr(), N, 0, 0, 0),
                       "gpfit_dgemm")
            return C
    A, B = _cu(A), _cu(B)
    va, vb = A.dim() == 1, B.dim() == 1
    if va:
        A = A[None, :] if not transA else A[:, None]
    if vb:
        B = B[:, None] if not transB else B[None, :]
    M, K = (A.shape[1], A.shape[0]) if transA else A.shape
    K2, N = (B.shape[1], B.shape[0]) if transB else B.shape
    if K != K2:
        raise RuntimeError(f"matmul: inner dimensions differ ({K} vs {K2})")
    Kp, Mp, Np = _ceil(K, 16), _ceil(M, 2), _ceil(N, 2)
    Ap = _pad2(A, Kp, Mp) if transA else _pad2(A, Mp, Kp)
    Bp = _pad2(B, Np, Kp) if transB else _pad2(B, Kp, Np)
    Cp = torch.empty((Mp, Np), dtype=TORCH_DTYPE, device=A.device)
    _lib.check(lib.gpfit_dgemm(_stream(), 1 if transA else 0, 0 if transB else 1, Mp, Np, Kp, float(alpha),
                               Ap.data_ptr(), Ap.stride(0), Bp.data_ptr(), Bp.stride(0), 0.0, Cp.data_ptr(),
                               Cp.stride(0), 0, 0, 0), "gpfit_dgemm")
    C = Cp[:M, :N]
    if va and vb:
        return C.reshape(()).clone()
    if vb:
        return C[:, 0].contiguous()   # never hand out strided views: callers pass data_ptr() on
    if va:
        return C[0, :].contiguous()
    return C.contiguous()


def gemm_into(C, A, B, alpha=1.0, beta=0.0, transA=False, transB=False):
    """``C <- alpha op(A) op(B) + beta C`` in place (``gpfit_dgemm`` with its beta): 2-D float64 device tensors, contiguous,
    K a multiple of 16, M and N even.  For iterations that would otherwise allocate and combine (eigtop's sign iteration)."""
    lib = _lib.load()
    M, K = (A.shape[1], A.shape[0]) if transA else A.shape
    K2, N = (B.shape[1], B.shape[0]) if transB else B.shape
    if K != K2 or tuple(C.shape) != (M, N) or (K & 15) or (M & 1) or (N & 1):
        raise RuntimeError("gemm_into: shapes do not fit (K % 16, even M and N, C = [M, N])")
    _lib.check(lib.gpfit_dgemm(_stream(), 1 if transA else 0, 0 if transB else 1, M, N, K, float(alpha), A.data_ptr(),
                               A.stride(0), B.data_ptr(), B.stride(0), float(beta), C.data_ptr(), C.stride(0), 0, 0, 0),
               "gpfit_dgemm")
    return C


def cholesky(M, want_inverse=False):
    """Lower Cholesky factor (and optionally its inverse) by the recursive MFMA algorithm.
    Returns ``(L, Linv_or_None, logdet, info)``; info > 0 = first non-positive pivot (LAPACK)."""
    lib = _lib.load()
    M = _cu(M)
    n = M.shape[0]
    eng = get_engine(n, 1)
    L = torch.empty((n, n), dtype=TORCH_DTYPE, device=M.device)
    Li = torch.empty((n, n), dtype=TORCH_DTYPE, device=M.device) if want_inverse else None
    logdet = ctypes.c_double()
    info = ctypes.c_int()
    rc = lib.gpfit_potrf(eng._ctx, _stream(), M.data_ptr(), M.stride(0), n, L.data_ptr(), L.stride(0),
                         Li.data_ptr() if want_inverse else None, Li.stride(0) if want_inverse else 0,
                         ctypes.byref(logdet), ctypes.byref(info))
    if rc < 0:
        _lib.check(rc, "gpfit_potrf")
    return L, Li, float(logdet.value), int(info.value)


def cholesky_append(L, Linv, kcol, logdet=0.0):
    """Factor of ``[[K, k], [k^T, kappa]]`` from the factor of K in O(n^2) (``gpfit_potrf_append``):
    ``L``, ``Linv`` are (n+1) x (n+1) device tensors whose leading n x n blocks hold the factor of K and
    its inverse (what ``cholesky(K, want_inverse=True)`` returned, embedded); ``kcol`` the n+1 entries of
    the new last column.  Row n of both is written in place; returns ``(logdet_new, info)``.  The closed
    loop of the reference's one_cell_active_training.ipynb adds one stimulus per iteration
    (:1889-1891): with this the refit's factorisation of K~ costs a row, not an N^3 / 3."""
    lib = _lib.load()
    n = L.shape[0] - 1
    if L.shape != (n + 1, n + 1) or Linv.shape != (n + 1, n + 1) or not (L.is_cuda and Linv.is_cuda):
        raise ValueError("cholesky_append: L and Linv must be (n+1) x (n+1) device tensors")
    if L.dtype != TORCH_DTYPE or Linv.dtype != TORCH_DTYPE or L.stride(1) != 1 or Linv.stride(1) != 1:
        raise ValueError("cholesky_append: L and Linv must be row-major float64")
    kcol = _cu(kcol).reshape(-1)
    if kcol.shape[0] != n + 1:
        raise ValueError("cholesky_append: kcol must hold n + 1 entries")
    eng = get_engine(n + 1, 1)
    ld = ctypes.c_double(float(logdet))
    info = ctypes.c_int()
    rc = lib.gpfit_potrf_append(eng._ctx, _stream(), L.data_ptr(), L.stride(0), Linv.data_ptr(), Linv.stride(0), n,
                                kcol.data_ptr(), ctypes.byref(ld), ctypes.byref(info))
    if rc < 0:
        _lib.check(rc, "gpfit_potrf_append")
    return float(ld.value), int(info.value)


def spd_inverse(M):
    """M^-1 = L^-T L^-1 for a symmetric positive definite matrix (replaces the LU
    ``torch.linalg.solve(M, I)`` of utils.py:2067)."""
    L, Li, _, info = cholesky(M, want_inverse=True)
    if info != 0:
        raise torch.linalg.LinAlgError(f"spd_inverse: matrix is not positive definite (info={info})")
    return matmul(Li, Li, transA=True)


# ------------------------------------------------------------------ numeric guards (utils.py:633-685)
def is_simmetric(tensor, name='M'):
    difference = (tensor - tensor.T).abs()
    if bool(torch.any(difference > MIN_TOLERANCE)):
        print(f'Matrix {name} is not symmetric, maximum difference is {difference.max()}')
        return False
    return True


def is_posdef(tensor, name='M'):
    if not is_simmetric(tensor, name=name):
        warnings.warn('The matrix is not symmetric, cannot check if it is positive definite')
        return False
    # cheap proof first: a successful Cholesky with 1 / (||L^-1||_1 ||L^-1||_inf) > MIN_TOLERANCE bounds the
    # smallest eigenvalue from below (see _all_eigenvalues_kept); eigvalsh only when that is inconclusive
    t = _cu(tensor)
    L, Li, _, info = cholesky(t, want_inverse=True)
    if info == 0:
        ali = Li.abs()
        if 1.0 / (float(ali.sum(0).max()) * float(ali.sum(1).max())) > MIN_TOLERANCE:
            return True
    smallest = float(torch.linalg.eigvalsh(t).min())
    if smallest <= 0.:
        warnings.warn(f'Matrix {name} is simmetric but has an eigenvalue smaller than 0 ')
        return False
    if smallest <= MIN_TOLERANCE:
        warnings.warn(f'Matrix {name} is simmetric but has an eigenvalue smaller than MIN_TOLERANCE: {MIN_TOLERANCE}')
        return False
    return True


def safe_log(x):
    if bool(torch.any(x <= 0)):
        raise ValueError("Negative or zero input to log detected")
    if bool(torch.any(x < 1e-10)):
        raise ValueError("Very small input to log detected")
    return torch.log(x)


def safe_acos(x):
    if bool(torch.any(x > 1 - 1e-6)) or bool(torch.any(x < -1 + 1e-6)):
        x = torch.clamp(x, -1 + 1e-6, 1 - 1e-6)
    return torch.acos(x)


def log_det(M, name='M', ignore_warning=False):
    """log|M| from the Cholesky factor, with the reference's fallbacks (utils.py:1271-1304):
    failed factorisation + symmetric -> sum of log of the eigenvalues above the truncation
    rule (with warnings); not symmetric -> warning and 0."""
    M = _cu(M)
    L, _, logdet, info = cholesky(M)
    if info == 0:
        safe_log(torch.diagonal(L))  # raises exactly when the reference's safe_log would
        return torch.tensor(logdet, dtype=TORCH_DTYPE, device=M.device)
    if is_simmetric(M, name=name):
        eigenvalues = torch.linalg.eigvalsh(M)
        ikeep = eigenvalues > max(float(eigenvalues.max()) * EIGVAL_TOL, EIGVAL_TOL)
        if not ignore_warning:
            smallest = float(eigenvalues.min())
            warnings.warn(f"Matrix {name} in logdet is simmetric but not posdef, using eigendecomposition to calculate the log determinant")
            if smallest <= 0.:
                warnings.warn(f'Matrix {name} in logdet is simmetric but has an eigenvalue smaller than 0 ')
            elif smallest <= 1.e-10:
                warnings.warn(f'Matrix {name} in logdet is simmetric but has an eigenvalue smaller than 1e-10 ')
        return torch.sum(safe_log(eigenvalues[ikeep]))
    warnings.warn(f"Matrix {name} in logdet is not simmetric in log_det used in KL_divergence")
    return 0


# ------------------------------------------------------------------ moments / likelihood / KL
def lambda_moments(x, K_tilde, KKtilde_inv, Kvec, K, C, m, V, theta, kernfun=None, dK=None, dK_tilde=None,
                   dK_vec=None, K_tilde_inv=None):
    """Mean and variance of lambda at the training points and, optionally, their
    theta-gradients (reference utils.py:1072-1124); every product on the MFMA GEMM."""
    a = _cu(KKtilde_inv)
    K, m, V = _cu(K), _cu(m), _cu(V)
    lambda_m = matmul(a, m)                                                       # :1090
    if Kvec is None:
        Kvec = kernfun(theta, x, x2=None, C=C, dC=None, diag=True)                # :1094
    aV = matmul(a, V)
    lambda_var = _cu(Kvec) + torch.sum(-K * a + a * aV, 1)                        # :1101 (V symmetric)
    if dK is None or dK_tilde is None or dK_vec is None or K_tilde_inv is None:
        return lambda_m, lambda_var
    dlambda_m, dlambda_var = {}, {}
    Kinv = _cu(K_tilde_inv)
    for key in dK.keys():
        da = matmul(_cu(dK[key]) - matmul(a, dK_tilde[key]), Kinv)                # :1114
        dlambda_m[key] = matmul(da, m)                                            # :1117
        dlambda_var[key] = (_cu(dK_vec[key]) + 2 * torch.sum(da * aV, 1) - torch.sum(_cu(dK[key]) * a, 1)
                            - torch.sum(K * da, 1))                               # :1120
    return lambda_m, lambda_var, dlambda_m, dlambda_var


def _lambda0_of(f_params):
    return torch.exp(f_params['loglambda0']) if 'loglambda0' in f_params else f_params['lambda0']


def _fparam_eval(lambda_m, lambda_var, r, logA, closed_form, lambda0=0.0, want_f=True):
    lib = _lib.load()
    lm, lv = _cu(lambda_m), _cu(lambda_var)
    n = lm.shape[0]
    rr = _cu(r) if r is not None else torch.zeros(n, dtype=TORCH_DTYPE, device=lm.device)
    eng = get_engine(n, 1)
    f = torch.empty(n, dtype=TORCH_DTYPE, device=lm.device) if want_f else None
    out = (ctypes.c_double * 7)()
    _lib.check(lib.gpfit_fparam_eval(eng._ctx, _stream(), lm.data_ptr(), lv.data_ptr(), rr.data_ptr(), n,
                                     _scalar(logA), 1 if closed_form else 0, float(lambda0),
                                     f.data_ptr() if want_f else None, out), "gpfit_fparam_eval")
    return f, list(out)


def mean_f_given_lambda_moments(f_params, lambda_m, lambda_var):
    """<f> = exp(A <lambda> + A^2/2 Var(lambda) + lambda0)   (utils.py:1126-1141)."""
    f, _ = _fparam_eval(lambda_m, lambda_var, None, f_params['logA'], False, _scalar(_lambda0_of(f_params)))
    return f


def lambda0_given_logA(logA, r, lambda_m, lambda_var):
    """Closed-form optimum of lambda0 given A (utils.py:1215-1229)."""
    _, out = _fparam_eval(lambda_m, lambda_var, r, logA, True, want_f=False)
    return torch.tensor(out[0], dtype=TORCH_DTYPE)


def mean_f(f_params, calculate_moments, lambda_m=None, lambda_var=None, x=None, K_tilde=None, KKtilde_inv=None,
           Kvec=None, K=None, C=None, m=None, V=None, V_inv=None, theta=None, kernfun=None, dK=None, dK_tilde=None,
           dK_vec=None, K_tilde_inv=None, r=None):
    """utils.py:1143-1213: firing-rate mean, computing the lambda moments first when asked to."""
    def rate(lm, lv):
        if r is not None:
            tmp = {'logA': f_params['logA'], 'lambda0': lambda0_given_logA(f_params['logA'], r, lm, lv)}
            return mean_f_given_lambda_moments(tmp, lm, lv)
        return mean_f_given_lambda_moments(f_params, lm, lv)

    if calculate_moments and (lambda_m is None or lambda_var is None):
        if dK is not None and dK_tilde is not None and dK_vec is not None and K_tilde_inv is not None:
            lambda_m, lambda_var, dlm, dlv = lambda_moments(x, K_tilde, KKtilde_inv, Kvec, K, C, m, V, theta,
                                                            kernfun=kernfun, dK=dK, dK_tilde=dK_tilde, dK_vec=dK_vec,
                                                            K_tilde_inv=K_tilde_inv)
            return rate(lambda_m, lambda_var), lambda_m, lambda_var, dlm, dlv
        lambda_m, lambda_var = lambda_moments(x, K_tilde, KKtilde_inv, Kvec, K, C, m, V, theta, kernfun=kernfun)
        return rate(lambda_m, lambda_var), lambda_m, lambda_var
    return rate(lambda_m, lambda_var)


def compute_loglikelihood(r, f_mean, lambda_m, lambda_var, f_params, compute_grad_for_f_params=False,
                          dlambda_m=None, dlambda_var=None):
    """utils.py:1231-1269.  O(N) reductions of device vectors (host-side glue)."""
    r, f_mean, lambda_m, lambda_var = _cu(r), _cu(f_mean), _cu(lambda_m), _cu(lambda_var)
    A = math.exp(_scalar(f_params['logA']))
    lambda0 = _scalar(_lambda0_of(f_params))
    rlambda_m = torch.dot(r, lambda_m)
    sum_r = torch.sum(r)
    loglikelihood = A * rlambda_m + lambda0 * sum_r - torch.sum(f_mean)           # :1243
    if compute_grad_for_f_params:
        d = {'logA': A * (rlambda_m - torch.dot(lambda_m + A * lambda_var, f_mean))}   # :1253
        if 'loglambda0' in f_params:
            d['loglambda0'] = (sum_r - torch.sum(f_mean)) * lambda0               # :1254
        elif 'lambda0' in f_params:
            d['lambda0'] = sum_r - torch.sum(f_mean)                              # :1255
        return loglikelihood, d
    if dlambda_m is not None and dlambda_var is not None:
        d = {}
        for key in dlambda_m.keys():
            d[key] = (A * torch.dot(r, dlambda_m[key]) - A * torch.dot(f_mean, dlambda_m[key])
                      - 0.5 * A * A * torch.dot(f_mean, dlambda_var[key]))        # :1266
        return loglikelihood, d
    return loglikelihood, rlambda_m, sum_r


def compute_KL_div(m, V, K_tilde, K_tilde_inv, dK_tilde=None, ignore_warning=False):
    """KL(q || p) and its theta-gradients (utils.py:1306-1337); no -n/2 term (:1326)."""
    m, V, K_tilde, K_tilde_inv = _cu(m), _cu(V), _cu(K_tilde), _cu(K_tilde_inv)
    c = matmul(V, K_tilde_inv)                                                    # :1318
    b = matmul(K_tilde_inv, m)                                                    # :1320
    KL = (-0.5 * log_det(V, name='V', ignore_warning=ignore_warning) + 0.5 * log_det(K_tilde, name='K_tilde')
          + 0.5 * torch.dot(m, b) + 0.5 * torch.trace(c))                         # :1326
    if dK_tilde is None:
        return KL
    dKL = {}
    for key in dK_tilde.keys():
        Bk = matmul(dK_tilde[key], K_tilde_inv)                                   # :1331
        dKL[key] = 0.5 * torch.trace(Bk) - 0.5 * torch.sum(c * Bk.T) - 0.5 * torch.dot(b, matmul(Bk, m))  # :1333
    return KL, dKL


def Estep(r, KKtilde_inv, m, f_params, f_mean, K_tilde=None, K_tilde_inv=None, V=None, update_V_inv=False, alpha=1):
    """Newton update of (m, V) -- the supported branch of the reference (alpha = 1,
    update_V_inv=False; utils.py:1420-1439).  ``V = (I + K~ G)^-1 K~`` is evaluated in its
    symmetric form ``L (I + L^T G L)^-1 L^T`` (K~ = L L^T) with two MFMA Cholesky factorisations
    instead of the reference's LU solve.  The alpha != 1 / update_V_inv branches are documented
    as not to be used (reference docs.md:5-7) and are not provided."""
    if update_V_inv or K_tilde is None or alpha != 1:
        warnings.warn('Estep: only the alpha = 1, update_V_inv = False update on V is implemented')
        raise NotImplementedError
    K_tilde = _cu(K_tilde)
    L, _, _, info = cholesky(K_tilde)
    if info != 0:
        raise torch.linalg.LinAlgError(f"Estep: K_tilde is not positive definite (info={info})")
    return _estep_given_factor(r, KKtilde_inv, m, f_params, f_mean, L)


def _estep_given_factor(r, KKtilde_inv, m, f_params, f_mean, L):
    """The Newton update of ``Estep`` with the Cholesky factor ``L`` of K~ (in the basis in force) supplied: K~ only
    changes with theta, so ``varGP`` factors it once per EM iteration, not once per E-step (utils.py:1864-1881 runs
    ``nEstep`` updates between two kernel rebuilds)."""
    r, a, m, f_mean = _cu(r), _cu(KKtilde_inv), _cu(m), _cu(f_mean)
    A = math.exp(_scalar(f_params['logA']))
    g = A * matmul(a, r - f_mean, transA=True)                                    # :1421
    G = A * A * matmul(a, a * f_mean[:, None], transA=True)                       # :1422
    n = L.shape[0]
    W = matmul(L, matmul(G, L), transA=True)
    W = (W + W.T) * 0.5 + torch.eye(n, dtype=TORCH_DTYPE, device=W.device)
    _, Lwi, _, info = cholesky(W, want_inverse=True)
    if info != 0:
        raise torch.linalg.LinAlgError(f"Estep: I + L^T G L is not positive definite (info={info})")
    P = matmul(L, Lwi, transB=True)
    V_new = matmul(P, P, transB=True)                                             # = solve(I + K~G, K~), :1430
    m_new = matmul(V_new, matmul(G, m) + g)                                       # :1431
    V_new = (V_new + V_new.T) / 2                                                 # :1438
    return m_new, V_new


def _estep_projected(r, KKtilde_inv, aL, L, m, f_params, f_mean, kv0=None):
    """``_estep_given_factor`` as ONE device call (``gpfit_estep_projected``: W = I + (diag(A sqrt f) a L)^T (...),
    its factor and inverse, V = L W^-1 L^T, m = L W^-1 (a L)^T u), with ``aL = a L`` supplied by the caller -- like
    ``L`` it only changes when the kernel is rebuilt.  A lab-shaped fit (3160 / 2100 images, 30 x 10 E-steps) spends
    a third of its time in these updates when they are issued product by product.  With ``kv0 = Kvec - rowsum(K o a)``
    the call also returns the moments of lambda behind the update (what ``lambda_moments`` would compute next)."""
    r, a, aL, L, m, f_mean = _cu(r), _cu(KKtilde_inv), _cu(aL), _cu(L), _cu(m), _cu(f_mean)
    a, aL, L = a.contiguous(), aL.contiguous(), L.contiguous()
    N, nb = a.shape
    m_new = torch.empty(nb, dtype=TORCH_DTYPE, device=a.device)
    V_new = torch.empty((nb, nb), dtype=TORCH_DTYPE, device=a.device)
    eng = get_engine(max(N, nb), 1)
    lam_m = lam_var = None
    if kv0 is not None:
        kv0 = _cu(kv0).contiguous()
        lam_m = torch.empty(N, dtype=TORCH_DTYPE, device=a.device)
        lam_var = torch.empty(N, dtype=TORCH_DTYPE, device=a.device)
    rc = _lib.load().gpfit_estep_projected(eng._ctx, _stream(), a.data_ptr(), a.stride(0), aL.data_ptr(), aL.stride(0),
                                           L.data_ptr(), L.stride(0), N, nb, r.contiguous().data_ptr(),
                                           m.contiguous().data_ptr(), f_mean.contiguous().data_ptr(),
                                           _scalar(f_params['logA']), m_new.data_ptr(), V_new.data_ptr(), V_new.stride(0),
                                           kv0.data_ptr() if kv0 is not None else None,
                                           lam_m.data_ptr() if kv0 is not None else None,
                                           lam_var.data_ptr() if kv0 is not None else None)
    if rc > 0:
        raise torch.linalg.LinAlgError(f"Estep: I + L^T G L is not positive definite (info={rc})")
    if rc != 0:
        raise _lib.GpfitError(f"gpfit_estep_projected: {_lib.last_error()} (rc={rc})")
    if kv0 is not None:
        return m_new, V_new, lam_m, lam_var
    return m_new, V_new


# ------------------------------------------------------------------ inference
def lambda_moments_star(xstar, xtilde, C, theta, K_tilde, K_tilde_inv, m, V, B, kernfun):
    """Predictive moments of lambda at test points (utils.py:1476-1500).  ``xstar`` may hold
    several rows: the reference's one-row-at-a-time loop (utils.py:388-397) is batched."""
    if kernfun == 'acosker':
        kernfun = acosker
    elif not callable(kernfun):
        raise Exception('Kernel function not recognized')
    Kvec_star = kernfun(theta, xstar, xtilde, C=C, dC=None, diag=False)          # :1486
    if not _is_identity(B):
        Kvec_star = matmul(Kvec_star, B)                                          # :1487
    a = matmul(Kvec_star, K_tilde_inv)                                            # :1489
    mu_star = matmul(a, m)                                                        # :1491
    K_star = kernfun(theta, xstar, x2=None, C=C, dC=None, diag=True)              # :1494
    sigma_star2 = K_star + torch.sum(matmul(a, _cu(V) - _cu(K_tilde)) * a, 1)     # :1498
    return mu_star, torch.reshape(sigma_star2, (_cu(xstar).reshape(-1, _cu(xstar).shape[-1]).shape[0],))


# ------------------------------------------------------------------ initialisation (utils.py:705-857)
def generate_xtilde(ntilde, x):
    """Jittered random subset of the first ``ntilde`` stimuli (utils.py:705-711)."""
    x = _cu(x)
    idx = torch.randperm(ntilde, device=x.device)
    first = x[idx, :]
    return first + torch.finfo(TORCH_DTYPE).eps * 10 * torch.randn(first.shape, dtype=TORCH_DTYPE, device=x.device)


def logbetaexpr_to_beta(logbetaexpr):
    return torch.exp(-0.5 * torch.as_tensor(logbetaexpr)) * torch.tensor(0.5)


def logrhoexpr_to_rho(logrhoexpr):
    return torch.exp(-0.5 * torch.as_tensor(logrhoexpr)) / torch.sqrt(torch.tensor(2.0))


def fromlogbetasam_to_logbetaexpr(logbetasam):
    return logbetasam - torch.log(torch.tensor(2.0))


def fromlogrhosam_to_logrhoexpr(logrhosam):
    return logrhosam - torch.log(torch.tensor(2.0))


def get_sta(x, r, n_px_side):
    """Spike-triggered average, its (hand-set) variance and the pixel of its peak (utils.py:736-753)."""
    x, r = _cu(x), _cu(r)
    n = r.shape[0]
    img_mean = matmul(x, torch.ones_like(r), transA=True) / n
    sta = matmul(x, r, transA=True) / n - img_mean
    rows, cols = _grid(n_px_side)
    peak = int(torch.argmax(torch.abs(sta)))
    return sta, torch.tensor(10), (torch.tensor(peak // cols), torch.tensor(peak % cols))


def generate_theta(x, r, n_px_side, display_hyper=False, **kwargs):
    """Initial hyperparameters and their boxes (utils.py:755-857).  As in the reference the
    receptive-field centre starts at (0, 0), the width is the hand-set 10 px^2, and caller
    overrides in ``kwargs`` are applied only when ``display_hyper`` is true (utils.py:829-834)."""
    up_lim, low_lim = 1, -1
    rows, _ = _grid(n_px_side)
    sigma_0 = torch.tensor(1.0, dtype=TORCH_DTYPE, requires_grad=True)
    Amp = torch.tensor(1.0, dtype=TORCH_DTYPE, requires_grad=True)
    get_sta(x, r, n_px_side)  # evaluated for parity of side effects; its peak is not used (utils.py:785-797)
    eps_0x = torch.tensor(0.0, dtype=TORCH_DTYPE, requires_grad=True)
    eps_0y = torch.tensor(0.0, dtype=TORCH_DTYPE, requires_grad=True)
    beta = (torch.sqrt(torch.tensor(10.0, dtype=TORCH_DTYPE)) / rows) * (up_lim - low_lim)
    logbetaexpr = (-2 * safe_log(2 * beta)).requires_grad_(True)
    rho = beta / 2
    logrhoexpr = (-safe_log(torch.tensor(2.0, dtype=TORCH_DTYPE) * (rho * rho))).requires_grad_(True)
    theta = {'sigma_0': sigma_0, 'eps_0x': eps_0x, 'eps_0y': eps_0y, '-2log2beta': logbetaexpr,
             '-log2rho2': logrhoexpr, 'Amp': Amp}
    if display_hyper:
        for key, value in kwargs.items():
            if key in theta:
                theta[key] = value
                print(f'updated {key} to {_scalar(value):.4f}')
        print(f' Dict of learnable hyperparameters : {", ".join(f"{k} = {_scalar(v):.4f}" for k, v in theta.items())}')
        print(f' Hyperparameters from the logexpr  : beta = {float(logbetaexpr_to_beta(logbetaexpr)):.4f}, '
              f'rho = {float(logrhoexpr_to_rho(logrhoexpr)):.4f}')
    inf = float('inf')
    lower = {'sigma_0': 0, 'eps_0x': low_lim, 'eps_0y': low_lim, '-2log2beta': -inf, '-log2rho2': -inf, 'Amp': 0.}
    upper = {'sigma_0': inf, 'eps_0x': up_lim, 'eps_0y': up_lim, '-2log2beta': inf, '-log2rho2': inf, 'Amp': inf}
    return (theta, lower, upper)


def print_hyp(theta):
    for key in theta.keys():
        extra = ''
        if key == '-2log2beta':
            extra = f' --> beta: {float(logbetaexpr_to_beta(theta[key])):8.4f}'
        if key == '-log2rho2':
            extra = f' --> rho : {float(logrhoexpr_to_rho(theta[key])):8.4f}'
        print(f' {key:<12}: {_scalar(theta[key]):>8.4f}{extra}')


# ------------------------------------------------------------------ active-learning utility (utils.py:413-525)
@torch.no_grad()
def nd_utility(sigma2, mu, r_masked):
    """Utility U = H(r|x,D) - <H(r|f,x)> of each candidate stimulus (reference utils.py:498-525
    with nd_p_r_given_xD / nd_lambda_r_mean / nd_mean_noise_entropy, 413-496).  ``sigma2`` and ``mu``
    are the variance and mean of log f for the candidates (0-d or [nstar]); ``r_masked`` the response
    counts of the truncated sum (the notebooks pass ``arange(0, 100)``).  One fused device kernel,
    Lambert W included (the reference goes through scipy on the host, utils.py:464-466)."""
    lib = _lib.load()
    s2, m = _cu(sigma2), _cu(mu)
    if s2.ndim == 0:                                     # utils.py:509-511
        s2, m = s2[None], m[None]
    s2, m = s2.reshape(-1).contiguous(), m.reshape(-1).contiguous()
    if s2.numel() != m.numel():
        raise ValueError("nd_utility: sigma2 and mu must have the same number of entries")
    r = _cu(r_masked).reshape(-1).contiguous()
    U = torch.empty_like(s2)
    _lib.check(lib.gpfit_nd_utility(_stream(), s2.data_ptr(), m.data_ptr(), s2.numel(), r.data_ptr(), r.numel(),
                                    U.data_ptr()), "gpfit_nd_utility")
    return U


# ------------------------------------------------------------------ metric (utils.py:1502-1541)
def explained_variance(rtst, f_pred, sigma=True):
    """Reliability-normalised r^2 between predicted rates and repeated test responses
    (rtst[repetitions, images], utils.py:1502-1541); with ``sigma`` the mean and spread over 1000
    random even/odd splits of the repetitions.  Reporting only (torch RNG, excluded from parity,
    SURVEY 8c G5).  The reference draws and evaluates the 1000 splits one at a time (a few
    thousand tiny kernels on a device: 0.4 s); here they are drawn and evaluated as one batch."""
    rtst, f_pred = _cu(rtst), _cu(f_pred)

    def corr(a, b):                       # Pearson correlation along the last axis (torch.corrcoef)
        a = a - a.mean(-1, keepdim=True)
        b = b - b.mean(-1, keepdim=True)
        return (a * b).sum(-1) / torch.sqrt((a * a).sum(-1) * (b * b).sum(-1))

    def r2_of(reven, rodd):
        rel = torch.abs(corr(reven, rodd))
        return 0.5 * (corr(f_pred, rodd) + corr(f_pred, reven)) / rel

    if not sigma:
        return r2_of(torch.mean(rtst[0::2, :], 0), torch.mean(rtst[1::2, :], 0)), None
    nboot, n = 1000, rtst.shape[0]
    perm = torch.argsort(torch.rand(nboot, n, device=rtst.device), dim=1)      # nboot random permutations
    reven = rtst[perm[:, 0::2]].mean(1)                                         # [nboot, images]
    rodd = rtst[perm[:, 1::2]].mean(1)
    vals = r2_of(reven, rodd)
    return torch.mean(vals), torch.std(vals)


# ------------------------------------------------------------------ fit (utils.py:1568-2316)
def _eigen_stabilise(K_tilde):
    """Spectral truncation of the reference (utils.py:1682-1683): eigh, keep
    lambda > max(lambda_max * EIGVAL_TOL, EIGVAL_TOL).  torch.linalg.eigh is host plumbing here
    (rank decision + basis), not part of the timed path."""
    eigvals, eigvecs = torch.linalg.eigh(K_tilde, UPLO='L')
    ikeep = eigvals > max(float(eigvals.max()) * EIGVAL_TOL, EIGVAL_TOL)
    return eigvals, eigvecs, ikeep


def _mark_identity(B):
    B._gpfit_identity = True       # fast-path hint only: products with B are skipped where it is seen
    return B


def _is_identity(B):
    return bool(getattr(B, '_gpfit_identity', False))


def _all_eigenvalues_kept(K_tilde):
    """Rank decision without an eigendecomposition, for the regime the reference's
    truncation rule (utils.py:1683, 1809) keeps EVERY eigenvalue: lambda_min > max(lambda_max tol, tol).

    From the Cholesky factorisation the fit needs anyway (K~ = L L^T, L^-1 by the MFMA recursion):
        lambda_max <= min(trace K~, ||K~||_inf)                 (SPD; Gershgorin)
        lambda_min  = 1 / ||L^-T L^-1||_2 >= 1 / (||L^-1||_1 ||L^-1||_inf)
    Both bounds are rigorous, so ``True`` is a proof that nothing would be truncated; when they are
    inconclusive (or K~ is not numerically positive definite) the caller falls back to
    ``torch.linalg.eigh``.  Returns ``(decided_all_kept, L, Linv)``."""
    L, Li, _, info = cholesky(K_tilde, want_inverse=True)
    if info != 0:
        return False, None, None
    return _all_kept_given_factor(K_tilde, Li), L, Li


def _all_kept_given_factor(K_tilde, Li):
    """The two rigorous bounds of ``_all_eigenvalues_kept`` for a K~ whose inverse factor ``Li`` is already known
    (the closed loop extends the factor by one row per added image, ``cholesky_append``)."""
    lam_max_ub = min(float(torch.diagonal(K_tilde).sum()), float(K_tilde.abs().sum(1).max()))
    ali = Li.abs()
    lam_min_lb = 1.0 / (float(ali.sum(0).max()) * float(ali.sum(1).max()))
    del ali
    if not (math.isfinite(lam_max_ub) and math.isfinite(lam_min_lb)):
        return False
    return lam_min_lb > max(lam_max_ub * EIGVAL_TOL, EIGVAL_TOL)


def _stabilised_basis(K_tilde, route=None, start=None):
    """Basis the reference works in after its eigen-stabilisation (utils.py:1682-1694): returns
    ``(eigvecs, B, K_tilde_b, K_tilde_inv_b)``.

    When every eigenvalue is provably kept (``_all_eigenvalues_kept``) B is square orthogonal and
    everything the reference computes downstream is basis-invariant, so the
    identity is used: ``B = I``, ``K_tilde_b = K~``, ``K_tilde_inv_b = L^-T L^-1`` -- no ``eigh``
    (0.67 s at N = 8192, five of them in a four-iteration fit).  When eigenvalues are truncated and
    N >= 256, the kept eigenspace comes from ``eigtop`` (block subspace iteration from N = 1792, the spectral projector of K~
    itself below) and the
    first return value holds only those columns; otherwise, and whenever that solver declines, the
    reference's own eigendecomposition + truncation.  Every route is a deterministic function of K~, so
    ``test(at_iteration=...)`` rebuilds the basis the tracked ``(m_b, V_b)`` were expressed in.

    The route taken (``'identity'``, ``'eigtop'`` or ``'eigh'``) is left in ``_BASIS.route`` (per host thread);
    ``varGP`` records it with every tracked iteration.  ``route=...`` asks for exactly that route -- what
    ``test(at_iteration=...)`` does with the recorded one, so that a model fitted under one setting
    (``GPFIT_FORCE_EIGH``, another library version) is never evaluated in a different basis: a route that
    cannot be reproduced raises instead of returning numbers in the wrong coordinates.

    ``start``: the solver state a previous call left in ``_BASIS.state`` for a NEARBY kernel matrix (``varGP`` hands the
    state of one EM iteration to the next: theta has moved a little, the kept eigenspace with it): the subspace solver
    then starts from that block and plans only the sweeps the measured distance needs.  Same stopping criterion; a call
    without it is a function of K~ alone (what ``test(at_iteration=...)`` relies on)."""
    n = K_tilde.shape[0]
    _BASIS.state = None

    def truncated_basis(want=None):
        # truncated regime: only the kept eigenspace, by block subspace iteration on the library's GEMM and Cholesky
        # (eigtop.py; 40 ms against 670 ms for the full eigh at N = 8192).  want = "subspace": the canonical
        # orthonormal basis of that space and the dense K~_b = B^T K~ B, no dense eigendecomposition at all;
        # "eigtop": its eigenvectors (one k x k eigh, k = 1024), same eigenvalues to 1e-14 and the same invariant
        # subspace to 1e-13 as the full eigh.  None: inconclusive -> the reference's own eigh.
        want = want or ("subspace" if EIGTOP_BASIS == "subspace" else "eigtop")
        if want == "subspace" and _DENSE_MIN_N <= n < _EIGTOP_MIN_N:
            # small matrices: the spectral projector of K~ itself (Cayley transform + scaled sign iteration on the
            # n x n matrix, no sweeps): 2.6 / 3.8 / 5.6 / 7.7 ms at n = 512 / 1024 / 1280 / 1536 against 12 / 22 / 29 /
            # 35 ms for rocSOLVER's eigh; the same canonical basis as the sweeps route gives for that space
            top = eigtop.kept_eigenspace_dense(K_tilde, EIGVAL_TOL, matmul, cholesky, gemm_into=gemm_into)
            if top is None:
                return None
            if top[1] is None:
                return "all-kept", None      # the projector is the identity: an exact all-kept proof (identity route)
            return "subspace", (top[1], top[1], top[2]["K_tilde_b"], top[2]["K_tilde_inv_b"])
        if n < _EIGTOP_MIN_N:
            return None
        top = eigtop.top_eigenpairs(K_tilde, EIGVAL_TOL, matmul, cholesky,
                                    basis="subspace" if want == "subspace" else "eigenvectors", gemm_into=gemm_into,
                                    start=start if want == "subspace" else None)
        if top is None:
            return None
        vals, vecs, info = top
        if vals is None:
            _BASIS.state = info.get("state")
            return "subspace", (vecs, vecs, info["K_tilde_b"], info["K_tilde_inv_b"])
        # the first entry stands in for the reference's N x N eigenvector matrix: only the kept columns exist
        return "eigtop", (vecs, vecs, torch.diag(vals), torch.diag_embed(1 / vals))

    def identity_basis(Li):
        B = _mark_identity(torch.eye(n, dtype=TORCH_DTYPE, device=K_tilde.device))
        Kinv = matmul(Li, Li, transA=True)
        return None, B, K_tilde, (Kinv + Kinv.T) * 0.5

    def eigh_basis():
        eigvals, eigvecs, ikeep = _eigen_stabilise(K_tilde)
        kept = eigvals[ikeep]
        return eigvecs, eigvecs[:, ikeep].contiguous(), torch.diag(kept), torch.diag_embed(1 / kept)

    if route is not None:
        if route == "identity":
            kept, L, Li = _all_eigenvalues_kept(K_tilde)
            if not kept and L is not None:
                got = truncated_basis("subspace")     # small matrices: the exact count where the norm bounds are not enough
                kept = got is not None and got[0] == "all-kept"
            out = identity_basis(Li) if kept else None
            if kept:
                _BASIS.factor = (L, Li)
        elif route in ("eigtop", "subspace"):
            got = truncated_basis(route)
            out = got[1] if (got is not None and got[0] == route) else None
        elif route == "eigh":
            out = eigh_basis()
        else:
            raise ValueError(f"unknown basis route {route!r}")
        if out is None:
            raise _lib.GpfitError(f"the basis route {route!r} recorded with this model cannot be reproduced for this "
                                  "kernel matrix (EIGVAL_TOL or the library changed since the fit)")
        _BASIS.route = route
        return out
    if not _FORCE_EIGH:
        # The two checks agree on every K~ (a proof that all eigenvalues are kept excludes a truncated count and
        # vice versa), so their order only decides what is paid: a kernel matrix of this size that was truncated
        # last time ON THIS THREAD (the same fit, one EM iteration later) goes to the subspace solver first and
        # skips the Cholesky + inverse + norms of the all-kept proof (~25 ms at N = 8192).
        hints = _BASIS.__dict__.setdefault("regime", {})
        key = (n, float(EIGVAL_TOL))
        exact_all = False        # the small-matrix projector found every eigenvalue above the threshold
        if hints.get(key) == "truncated":
            got = truncated_basis()
            if got is not None and got[0] != "all-kept":
                _BASIS.route = got[0]
                return got[1]
            exact_all = got is not None
        kept, L, Li = _all_eigenvalues_kept(K_tilde)
        if not kept and not exact_all and hints.get(key) != "truncated":
            got = truncated_basis()
            if got is not None and got[0] != "all-kept":
                hints[key] = "truncated"
                _BASIS.route = got[0]
                return got[1]
            exact_all = got is not None
        if (kept or exact_all) and L is not None:
            # proved by the norm bounds of _all_eigenvalues_kept or, where those are inconclusive on a small matrix
            # (BASELINE configs[0]: lambda_min within 2 % of the threshold), by the exact count of the spectral projector
            hints[key] = "full"
            _BASIS.route = "identity"
            _BASIS.factor = (L, Li)      # K~ = L L^T, L^-1: kept by varGP for the closed loop (extend_inducing_set)
            return identity_basis(Li)
    _BASIS.route = "eigh"
    return eigh_basis()


import os as _os_mod
_FORCE_EIGH = bool(_os_mod.environ.get("GPFIT_FORCE_EIGH"))   # tuning / A-B knob: always take the eigh route
_EIGTOP_MIN_N = 1792   # from here up the kept eigenspace comes from block subspace sweeps (warm-started: 10.4 ms at N = 1792,
                       # 10.8 at 2048, 11.7 at 2560); below, down to _DENSE_MIN_N, from the spectral projector of K~ itself
                       # (no sweeps: 9.6 ms at N = 1792, 11.4 at 2048, 22 at 2560) -- profiles/r04_small_n_basis.log.  The
                       # kept count is 530-580 whatever N is, so below ~1800 a block iteration has nothing to discard.
_DENSE_MIN_N = 256     # below this the reference's own eigh (a few ms)
# What the truncated regime's basis B is made of at N >= 256 (module global read at call time, like EIGVAL_TOL):
#   "subspace"     (default) the canonical orthonormal basis of the kept EIGENSPACE: no dense eigendecomposition at
#                  all; K_tilde_b = B^T K~ B is a dense n x n matrix.  Everything downstream is invariant under the
#                  choice of an orthonormal basis of that space (the reference's own formulas only use B^T B = I and the
#                  invariance of span B), but the COLUMNS of B are not eigenvectors;
#   "eigenvectors" the reference's own choice (utils.py:1683-1694): the kept eigenvectors, K_tilde_b diagonal, at the
#                  price of one k x k eigendecomposition (k = 1024) per basis: 16 ms more per EM iteration.
EIGTOP_BASIS = _os_mod.environ.get("GPFIT_EIGTOP_BASIS", "subspace")
# per host thread (the reference's active-learning notebook fits and scores on two threads): the route the last
# call of _stabilised_basis took, and (N, EIGVAL_TOL) -> "full" | "truncated", which of the two rank checks to try
# first -- a hint only, so one thread's history never changes what another thread's fit costs
_BASIS = threading.local()


def _closure_general(theta, lims, n_px_side, x, xtilde, r, B, m_b, V_b, f_params, ntilde, nt):
    """The M-step closure body in the reference's own B-projected formulation
    (utils.py:2030-2099) on the GPU primitives: used when eigenvalues were truncated
    (n < n_tilde) or n_tilde != n_t, where the original-basis fast path does not apply."""
    lower, upper = lims
    C, mask, dC = localker(theta=theta, theta_higher_lims=upper, theta_lower_lims=lower, n_px_side=n_px_side, grad=True)
    xt_m = xtilde[:, mask].contiguous()
    K_tilde, dK_tilde = acosker(theta, xt_m, xt_m, C=C, dC=dC, diag=False)
    if ntilde != nt:
        x_m = x[:, mask].contiguous()
        K, dK = acosker(theta, x_m, xt_m, C=C, dC=dC, diag=False)
    else:
        x_m = xt_m if x is xtilde else x[:, mask].contiguous()
        K, dK = K_tilde, dK_tilde
    Kvec, dKvec = acosker(theta, x_m, x2=None, C=C, dC=dC, diag=True)
    K_tilde_b = matmul(B, matmul(K_tilde, B), transA=True)                         # :2047
    K_tilde_b = (K_tilde_b + K_tilde_b.T) * 0.5                                    # :2048
    K_b = matmul(K, B)                                                             # :2049
    dK_tilde_b = {k: matmul(B, matmul(dK_tilde[k], B), transA=True) for k in dK_tilde}   # :2061
    dK_b = {k: matmul(dK[k], B) for k in dK}                                       # :2062
    K_tilde_inv_b = spd_inverse(K_tilde_b)                                         # :2067 (Cholesky instead of LU)
    KKtilde_inv_b = matmul(K_b, K_tilde_inv_b) if ntilde != nt else B              # :2068
    f_mean, lambda_m, lambda_var, dlm, dlv = mean_f(
        f_params=f_params, calculate_moments=True, x=x_m, K_tilde=K_tilde_b, KKtilde_inv=KKtilde_inv_b, Kvec=Kvec,
        K=K_b, C=C, m=m_b, V=V_b, theta=theta, kernfun=acosker, dK=dK_b, dK_tilde=dK_tilde_b, dK_vec=dKvec,
        K_tilde_inv=K_tilde_inv_b)                                                 # :2070
    loglik, dloglik = compute_loglikelihood(r, f_mean, lambda_m, lambda_var, f_params, dlambda_m=dlm, dlambda_var=dlv)
    KL, dKL = compute_KL_div(m_b, V_b, K_tilde_b, K_tilde_inv=K_tilde_inv_b, dK_tilde=dK_tilde_b)
    grad = {k: -(_scalar(dloglik[k]) - _scalar(dKL[k])) for k in theta.keys()}      # :2097-2099
    return -(_scalar(loglik) - _scalar(KL)), grad                                  # :2087-2089


def _closure_projected(theta, lims, n_px_side, x, r, B, m_b, V_b, f_params):
    """Truncated-rank M-step closure (utils.py:2030-2099 with n < n_tilde = n_t): ONE call of the fused
    entry point ``gpfit_fit_eval_projected`` (kernel build, projection, the two n x n Cholesky
    factorisations, moments / likelihood / KL, adjoints, lift and pull-back on the device).  When a
    factorisation meets a non-positive pivot the reference's ``log_det`` would take its
    eigen-fallback (utils.py:1279-1304): the step-by-step formulation below (``_closure_projected_steps``)
    reproduces that and is used instead."""
    lib = _lib.load()
    lower, upper = lims
    xc, rc, Bc, mc, Vc = _cu(x), _cu(r), _cu(B), _cu(m_b), _cu(V_b)
    rows, cols = _grid(n_px_side)
    N, nk = Bc.shape
    eng = get_engine(N, xc.shape[1], rows * cols)
    out = (ctypes.c_double * 16)()
    rc_ = lib.gpfit_fit_eval_projected(eng._ctx, _stream(), _lib.darr(theta_vec(theta)),
                                       _lib.darr([_scalar(lower[k]) for k in THETA_KEYS]),
                                       _lib.darr([_scalar(upper[k]) for k in THETA_KEYS]), rows, cols,
                                       xc.data_ptr(), xc.stride(0), N, rc.data_ptr(), Bc.data_ptr(), Bc.stride(0), nk,
                                       mc.data_ptr(), Vc.data_ptr(), Vc.stride(0), _scalar(f_params['logA']),
                                       _scalar(_lambda0_of(f_params)), out)
    if rc_ == -2:
        raise ValueError(_lib.last_error())
    if rc_ < 0:
        _lib.check(rc_, "gpfit_fit_eval_projected")
    if rc_ == 0:
        return out[0], {k: out[3 + i] for i, k in enumerate(THETA_KEYS)}
    return _closure_projected_steps(theta, lims, n_px_side, x, r, B, m_b, V_b, f_params)


def _closure_projected_steps(theta, lims, n_px_side, x, r, B, m_b, V_b, f_params):
    """Truncated-rank M-step closure (utils.py:2030-2099 with n < n_tilde = n_t, inducing set =
    training set) in adjoint form: the same loss as the reference's B-projected formulation, but
    instead of materialising the six dK~_p and projecting each of them (13 + 13 GEMMs of N x N x n),
    the n x n / N x n adjoints of the loss are formed once, lifted to
    ``W = (B G_Kb~ + G_Kb) B^T`` and contracted with the analytic dK~_p by the fused pull-back
    (``gpfit_grad_pullback``).  As in the reference, with n_tilde == n_t the moments use a = B
    (utils.py:2068) while their derivatives use da_p = (dK_b,p - a dK~_b,p) K~_b^-1 (utils.py:1114);
    collecting the coefficients of dK_b,p, dK~_b,p and dKvec_p in dKL_p - dloglik_p
    (utils.py:1117-1120, 1266, 1331-1333) with g_m = A (r - f), g_v = -A^2 f / 2:
      G_a   = g_m m_b^T - diag(g_v) K_b + 2 diag(g_v) B V_b
      G_Kb  = diag(g_v) B - G_a K~_b^-1
      G_Kb~ = 1/2 K~_b^-1 - 1/2 b b^T - 1/2 K~_b^-1 V_b K~_b^-1 + B^T G_a K~_b^-1 ,  gvec = -g_v.
    Validated against the reference on the truncated golden fixture (1e-10) and against the
    literal formulation (``_closure_general``) at scale."""
    lib = _lib.load()
    lower, upper = lims
    th = _lib.darr(theta_vec(theta))
    if lib.gpfit_check_limits(th, _lib.darr([_scalar(lower[k]) for k in THETA_KEYS]),
                              _lib.darr([_scalar(upper[k]) for k in THETA_KEYS])) != 0:
        raise ValueError(_lib.last_error())
    C, mask = localker(theta=theta, theta_higher_lims=upper, theta_lower_lims=lower, n_px_side=n_px_side, grad=False)
    x_m = x[:, mask].contiguous()
    K_tilde = acosker(theta, x_m, x_m, C=C, dC=None, diag=False)
    Kvec = acosker(theta, x_m, x2=None, C=C, dC=None, diag=True)
    A = math.exp(_scalar(f_params['logA']))
    lambda0 = _scalar(_lambda0_of(f_params))
    K_b = matmul(K_tilde, B)                                                        # :2049
    K_tilde_b = matmul(B, K_b, transA=True)                                         # :2047
    K_tilde_b = (K_tilde_b + K_tilde_b.T) * 0.5                                     # :2048
    Ki = spd_inverse(K_tilde_b)                                                     # :2067
    a = B                                                                           # :2068 (n_tilde == n_t)
    aV = matmul(a, V_b)
    lambda_m = matmul(a, m_b)                                                       # :1090
    lambda_var = Kvec - torch.sum(a * K_b, 1) + torch.sum(aV * a, 1)                 # :1101
    f_mean = torch.exp(A * lambda_m + 0.5 * A * A * lambda_var + lambda0)            # :1138
    loglik = A * torch.dot(r, lambda_m) + lambda0 * torch.sum(r) - torch.sum(f_mean)  # :1243
    b = matmul(Ki, m_b)
    KiV = matmul(Ki, V_b)
    KL = (-0.5 * log_det(V_b, 'V', ignore_warning=True) + 0.5 * log_det(K_tilde_b, 'K_tilde', ignore_warning=True)
          + 0.5 * torch.dot(m_b, b) + 0.5 * torch.trace(KiV))                        # :1326
    g_m = A * (r - f_mean)
    g_v = -0.5 * A * A * f_mean
    G_a = torch.outer(g_m, m_b) - g_v[:, None] * K_b + 2.0 * g_v[:, None] * aV
    G_aKi = matmul(G_a, Ki)
    G_Kb = g_v[:, None] * a - G_aKi
    G_Ktb = 0.5 * Ki - 0.5 * torch.outer(b, b) - 0.5 * matmul(KiV, Ki) + matmul(a, G_aKi, transA=True)
    W = matmul(matmul(B, G_Ktb) + G_Kb, B, transB=True)
    W = ((W + W.T) * 0.5).contiguous()
    gvec = (-g_v).contiguous()
    rows, cols = _grid(n_px_side)
    xc = _cu(x)
    eng = get_engine(xc.shape[0], int(mask.sum()), rows * cols)
    out = (ctypes.c_double * 6)()
    _lib.check(lib.gpfit_grad_pullback(eng._ctx, _stream(), th, rows, cols, xc.data_ptr(), xc.stride(0), xc.shape[0],
                                       W.data_ptr(), W.stride(0), gvec.data_ptr(), out), "gpfit_grad_pullback")
    grad = {k: out[i] for i, k in enumerate(THETA_KEYS)}
    return -(_scalar(loglik) - _scalar(KL)), grad


def _closure_sparse(theta, lims, n_px_side, x, xtilde, r, B, m_b, V_b, f_params):
    """Sparse M-step closure (n_tilde < n_t; utils.py:2030-2099, 1114-1120): ONE call of the fused entry
    point ``gpfit_fit_eval_sparse``; the step-by-step formulation below (``_closure_sparse_steps``) is
    the path taken when a factorisation meets a non-positive pivot (reference's eigen-fallback of
    ``log_det``, utils.py:1279-1304)."""
    lib = _lib.load()
    lower, upper = lims
    xc, xtc, rc, Bc, mc, Vc = _cu(x), _cu(xtilde), _cu(r), _cu(B), _cu(m_b), _cu(V_b)
    rows, cols = _grid(n_px_side)
    eng = get_engine(max(xc.shape[0], xtc.shape[0]), xc.shape[1], rows * cols)
    out = (ctypes.c_double * 16)()
    rc_ = lib.gpfit_fit_eval_sparse(eng._ctx, _stream(), _lib.darr(theta_vec(theta)),
                                    _lib.darr([_scalar(lower[k]) for k in THETA_KEYS]),
                                    _lib.darr([_scalar(upper[k]) for k in THETA_KEYS]), rows, cols,
                                    xc.data_ptr(), xc.stride(0), xc.shape[0], xtc.data_ptr(), xtc.stride(0), xtc.shape[0],
                                    rc.data_ptr(), Bc.data_ptr(), Bc.stride(0), Bc.shape[1], mc.data_ptr(), Vc.data_ptr(),
                                    Vc.stride(0), _scalar(f_params['logA']), _scalar(_lambda0_of(f_params)), out)
    if rc_ == -2:
        raise ValueError(_lib.last_error())
    if rc_ < 0:
        _lib.check(rc_, "gpfit_fit_eval_sparse")
    if rc_ == 0:
        return out[0], {k: out[3 + i] for i, k in enumerate(THETA_KEYS)}
    return _closure_sparse_steps(theta, lims, n_px_side, x, xtilde, r, B, m_b, V_b, f_params)


def _closure_sparse_steps(theta, lims, n_px_side, x, xtilde, r, B, m_b, V_b, f_params):
    """Sparse M-step closure (n_tilde < n_t: K[n_t, n_tilde] != K~, a = K_b K~_b^-1 with non-zero
    da_p; utils.py:2030-2099, 1114-1120) in adjoint form.  Two adjoint matrices come out of the
    same algebra as in ``_closure_projected`` (with a = K_b K~_b^-1 this time):
      W~ = B G_Kb~ B^T  (n_tilde x n_tilde, contracts with dK~_p)  -> ``gpfit_grad_pullback`` on xtilde,
      W_K = G_Kb B^T    (n_t x n_tilde, contracts with dK_p)       -> rectangular pull-back below:
    with c, delta the cosine / angle matrix of (x, xtilde), A_w = W_K o (pi - delta)/pi,
    B_m = W_K o sqrt(1 - c^2)/pi, u1 = B_m q2, u2 = B_m^T q1,
      sum_ij W_K,ij dK_p,ij = <dC_p, x^T A_w xt + x^T diag(u1 / 2 q1) x + xt^T diag(u2 / 2 q2) xt>
    (sigma_0: 2 s0 sum A_w + s0 sum u1/q1 + s0 sum u2/q2), from utils.py:996-1021; the dKvec term
    adds x^T diag(gvec) x.  Both pull-backs are fused entry points (``gpfit_grad_pullback``,
    ``gpfit_acosker_pullback``); only the five d x d contractions with dC_p stay in torch.
    Validated against the reference on the sparse golden fixture."""
    lib = _lib.load()
    lower, upper = lims
    th = _lib.darr(theta_vec(theta))
    C, mask, dC = localker(theta=theta, theta_higher_lims=upper, theta_lower_lims=lower, n_px_side=n_px_side, grad=True)
    x_m, xt_m = x[:, mask].contiguous(), xtilde[:, mask].contiguous()
    s0 = _scalar(theta['sigma_0'])
    K_tilde = acosker(theta, xt_m, xt_m, C=C, dC=None, diag=False)
    Kvec = acosker(theta, x_m, x2=None, C=C, dC=None, diag=True)
    K = acosker(theta, x_m, xt_m, C=C, dC=None, diag=False)                          # :968-990
    A = math.exp(_scalar(f_params['logA']))
    lambda0 = _scalar(_lambda0_of(f_params))
    K_b = matmul(K, B)                                                              # :2049
    K_tilde_b = matmul(B, matmul(K_tilde, B), transA=True)                          # :2047
    K_tilde_b = (K_tilde_b + K_tilde_b.T) * 0.5                                     # :2048
    Ki = spd_inverse(K_tilde_b)                                                     # :2067
    a = matmul(K_b, Ki)                                                             # :2068
    aV = matmul(a, V_b)
    lambda_m = matmul(a, m_b)                                                       # :1090
    lambda_var = Kvec - torch.sum(a * K_b, 1) + torch.sum(aV * a, 1)                 # :1101
    f_mean = torch.exp(A * lambda_m + 0.5 * A * A * lambda_var + lambda0)            # :1138
    loglik = A * torch.dot(r, lambda_m) + lambda0 * torch.sum(r) - torch.sum(f_mean)  # :1243
    b = matmul(Ki, m_b)
    KiV = matmul(Ki, V_b)
    KL = (-0.5 * log_det(V_b, 'V', ignore_warning=True) + 0.5 * log_det(K_tilde_b, 'K_tilde', ignore_warning=True)
          + 0.5 * torch.dot(m_b, b) + 0.5 * torch.trace(KiV))                        # :1326
    g_m = A * (r - f_mean)
    g_v = -0.5 * A * A * f_mean
    G_a = torch.outer(g_m, m_b) - g_v[:, None] * K_b + 2.0 * g_v[:, None] * aV
    G_aKi = matmul(G_a, Ki)
    G_Kb = g_v[:, None] * a - G_aKi
    G_Ktb = 0.5 * Ki - 0.5 * torch.outer(b, b) - 0.5 * matmul(KiV, Ki) + matmul(a, G_aKi, transA=True)
    # square part: W~ against dK~_p
    Wt = matmul(matmul(B, G_Ktb), B, transB=True)
    Wt = ((Wt + Wt.T) * 0.5).contiguous()
    rows, cols = _grid(n_px_side)
    xtc = _cu(xtilde)
    eng = get_engine(xtc.shape[0], int(mask.sum()), rows * cols)
    zero = torch.zeros(xtc.shape[0], dtype=TORCH_DTYPE, device=xtc.device)
    out = (ctypes.c_double * 6)()
    _lib.check(lib.gpfit_grad_pullback(eng._ctx, _stream(), th, rows, cols, xtc.data_ptr(), xtc.stride(0), xtc.shape[0],
                                       Wt.data_ptr(), Wt.stride(0), zero.data_ptr(), out), "gpfit_grad_pullback")
    grad = {k: out[i] for i, k in enumerate(THETA_KEYS)}
    # rectangular part: W_K against dK_p, and gvec against dKvec_p (fused pull-back, nothing n_t x n_tilde x 6)
    W_K = matmul(G_Kb, B, transB=True).contiguous()
    gvec = (-g_v).contiguous()
    d = int(mask.sum())
    Cc = _cu(C)
    M = torch.empty((d, d), dtype=TORCH_DTYPE, device=x_m.device)
    out3 = (ctypes.c_double * 3)()
    eng2 = get_engine(max(x_m.shape[0], xt_m.shape[0]), d, rows * cols)
    _lib.check(lib.gpfit_acosker_pullback(eng2._ctx, _stream(), s0, x_m.data_ptr(), x_m.stride(0), x_m.shape[0],
                                          xt_m.data_ptr(), xt_m.stride(0), xt_m.shape[0], d, Cc.data_ptr(), Cc.stride(0),
                                          W_K.data_ptr(), W_K.stride(0), gvec.data_ptr(), M.data_ptr(), M.stride(0), out3),
               "gpfit_acosker_pullback")
    M = (M + M.T) * 0.5
    for k in DC_KEYS:
        grad[k] += _scalar(torch.sum(dC[k] * M))
    grad['sigma_0'] += s0 * (2.0 * out3[0] + out3[1] + out3[2]) + 2 * s0 * _scalar(gvec.sum())
    return -(_scalar(loglik) - _scalar(KL)), grad


@torch.no_grad()
def extend_inducing_set(fit_model, x_new, route=None):
    """One image more in the inducing set of a fitted model: everything ``varGP(..., m=, V=, init_kernel=)`` needs
    to refit from the previous posterior, as the closed loop of ``one_cell_active_training.ipynb`` (:1889-1936)
    prepares it -- K~ grown "by its latest column", the stabilised basis of the grown matrix, the previous
    ``(m, V)`` carried over with a unit variance and the mean of ``m`` for the new point.

    Where the notebook eigendecomposes the grown K~ for every added image (:1900), this uses what the previous fit
    left behind: when its basis was the identity (every eigenvalue kept) and its factor is in
    ``final_kernel['chol']``, ``L`` and ``L^-1`` are extended by one row in O(n^2) (``cholesky_append``,
    ``gpfit_potrf_append``) and the same rigorous bounds that proved "all kept" for the old matrix are evaluated
    for the new one -- no ``eigh``, no refactorisation.  If the bounds no longer prove it (or no factor was kept)
    the grown matrix goes through ``_stabilised_basis`` like any other.

    Returns ``dict(xtilde=, m=, V=, init_kernel=)``; ``x_new`` is one image (any shape with the model's pixel
    count).  The caller updates ``fit_parameters['ntilde']``.  PRECONDITION (the closed loop of the notebook, where
    ``in_use_idx == xtilde_idx``): the refit's training set is the grown inducing set, ``x == xtilde`` -- the
    ``init_kernel`` returned here carries ``K = K~`` and ``KKtilde_inv_b = B``, which is only the kernel of THAT
    training set; ``varGP`` refuses it for any other (``init_kernel['square']``).
    ``route='eigh'`` forces the notebook's own step (eigendecomposition of the grown matrix), for A/B timing."""
    xt = _cu(fit_model['xtilde'])
    n, npx = xt.shape
    row = _cu(x_new).reshape(1, -1)
    if row.shape[1] != npx:
        raise ValueError(f"extend_inducing_set: the new image has {row.shape[1]} pixels, the model {npx}")
    theta = fit_model['hyperparams_tuple'][0]
    fk = fit_model['final_kernel']
    C, mask = _cu(fit_model['C']), fit_model['mask'].to(xt.device)
    K_old, Kvec_old = _cu(fk['K_tilde']), _cu(fk['Kvec'])
    xt_new = torch.cat((xt, row), 0)
    xm = xt_new[:, mask].contiguous()
    col = acosker(theta, xm, xm[n:n + 1].contiguous(), C=C, dC=None, diag=False).reshape(-1)       # latest column, n + 1 entries
    K_new = torch.empty((n + 1, n + 1), dtype=TORCH_DTYPE, device=xt.device)
    K_new[:n, :n] = K_old
    K_new[:n, n] = col[:n]
    K_new[n, :] = col
    Kvec_new = torch.cat((Kvec_old, acosker(theta, xm[n:n + 1].contiguous(), x2=None, C=C, dC=None, diag=True).reshape(-1)))
    # previous posterior in the coordinates of the images, grown by the new point (notebook: identity block, mean of m)
    B_old = fit_model['B']
    if _is_identity(B_old) or fit_model.get('basis_route') == 'identity':
        m_img, V_img = _cu(fit_model['m_b']), _cu(fit_model['V_b'])
    else:
        B_old = _cu(B_old)
        m_img = matmul(B_old, _cu(fit_model['m_b']))
        V_img = matmul(B_old, matmul(_cu(fit_model['V_b']), B_old, transB=True))
    V_new = torch.eye(n + 1, dtype=TORCH_DTYPE, device=xt.device)
    V_new[:n, :n] = (V_img + V_img.T) * 0.5
    m_new = torch.cat((m_img, m_img.mean().reshape(1)))
    # basis of the grown matrix
    forced, route, chol = route, None, None
    if forced is None and fit_model.get('basis_route') == 'identity' and fk.get('chol') is not None:
        L = torch.zeros((n + 1, n + 1), dtype=TORCH_DTYPE, device=xt.device)
        Li = torch.zeros_like(L)
        L[:n, :n] = _cu(fk['chol']['L'])
        Li[:n, :n] = _cu(fk['chol']['Linv'])
        _, info = cholesky_append(L, Li, col)
        if info == 0 and _all_kept_given_factor(K_new, Li):
            route, chol = 'identity', {'L': L, 'Linv': Li}
            B = _mark_identity(torch.eye(n + 1, dtype=TORCH_DTYPE, device=xt.device))
            Kinv = matmul(Li, Li, transA=True)
            K_b_, K_inv_b = K_new, (Kinv + Kinv.T) * 0.5
    if route is None:
        _BASIS.factor = None
        _, B, K_b_, K_inv_b = _stabilised_basis(K_new, route=forced)
        route = _BASIS.route
        if route == 'identity' and _BASIS.factor is not None:
            chol = {'L': _BASIS.factor[0], 'Linv': _BASIS.factor[1]}
        _BASIS.factor = None
    KB = K_new if route == 'identity' else matmul(K_new, B)
    init_kernel = {'C': C, 'mask': mask, 'K_tilde': K_new, 'K': K_new, 'Kvec': Kvec_new, 'B': B, 'K_tilde_b': K_b_, 'K_b': KB,
                   'K_tilde_inv_b': K_inv_b, 'KKtilde_inv_b': B, 'basis_route': route, 'chol': chol, 'square': True}
    return {'xtilde': xt_new, 'm': m_new, 'V': V_new, 'init_kernel': init_kernel}


@torch.no_grad()
def varGP(x, r, **kwargs):
    """Variational-GP fit (EM) with the reference's call signature and ``fit_model`` schema
    (utils.py:1568-2316): ``varGP(x, r, fit_parameters=..., xtilde=..., hyperparams_tuple=...,
    f_params=..., [m, V, init_kernel])`` -> ``(fit_model, err_dict)``.

    State is kept, as in the reference, in the eigenbasis ``B`` of K~ (``m_b``, ``V_b``).  While
    every eigenvalue is kept and the inducing set is the training set (the regime of the
    north-star configurations) the M-step closure is ONE call of the fused HIP unit of work
    (``gpfit_fit_eval``) and the E-step ONE call of ``gpfit_estep`` in the original basis;
    otherwise the same steps run in the reference's projected formulation on the GPU
    primitives.

    Deviations from the reference's ``fit_model``, both additive or opt-out:
    * ``final_kernel['eigvecs']`` (utils.py:2241: the N x N eigenvector matrix of K~) is ``None`` when every
      eigenvalue was kept (no eigendecomposition is computed on that route) and holds only the kept columns when
      the subspace solver built the basis (N >= 256, truncated spectrum); ``fit_parameters['full_eigvecs'] =
      True`` asks for the reference's full matrix (one ``torch.linalg.eigh`` of the final K~ at the end of the fit).
    * extra keys: ``fit_model['basis_route']`` and ``values_track['variation_par_track']['basis_route']``
      (``'identity'`` / ``'eigtop'`` / ``'eigh'`` per tracked iteration): the route that built the basis ``(m_b,
      V_b)`` are expressed in, which ``test(at_iteration=...)`` reproduces or refuses."""
    import time
    start_time_before_init = time.time()
    err_dict = {'is_error': False, 'error_message': None}
    x, r = _cu(x), _cu(r)
    nt, nx = x.shape
    dev = x.device
    reserve(nt, nx, nx)

    fit_parameters = copy.deepcopy(kwargs['fit_parameters'])
    fit_parameters['min_tolerance'] = MIN_TOLERANCE
    fit_parameters['eigval_tol'] = EIGVAL_TOL
    ntilde = fit_parameters.get('ntilde', 100 if nt > 100 else nt)
    maxiter = fit_parameters.get('maxiter', 50)
    nEstep = fit_parameters.get('nEstep', 50)
    nMstep = fit_parameters.get('nMstep', 20)
    nFparamstep = fit_parameters.get('nFparamstep', 10)
    display_hyper = fit_parameters.get('display_hyper', True)
    n_px_side = fit_parameters.get('n_px_side', math.sqrt(nx))
    if fit_parameters.get('kernfun', 'acosker') != 'acosker':
        raise Exception('Kernel function not recognized')
    kernfun = acosker

    x_given_as_xtilde = 'xtilde' in kwargs and kwargs['xtilde'] is x
    xtilde = _cu(kwargs['xtilde']) if 'xtilde' in kwargs else generate_xtilde(ntilde, x)
    if ntilde != xtilde.shape[0]:
        raise Exception('Number of inducing points does not match ntilde')
    hyperparams_tuple = (copy.deepcopy(kwargs['hyperparams_tuple']) if 'hyperparams_tuple' in kwargs
                         else generate_theta(x, r, n_px_side, display_hyper))
    theta = copy.deepcopy(kwargs.get('theta', hyperparams_tuple[0]))
    theta_lower_lims = copy.deepcopy(kwargs.get('theta_lower_lims', hyperparams_tuple[1]))
    theta_higher_lims = copy.deepcopy(kwargs.get('theta_higher_lims', hyperparams_tuple[2]))
    lims = (theta_lower_lims, theta_higher_lims)
    if 'f_params' not in kwargs:
        raise Exception('f_params not provided')
    f_params = copy.deepcopy(kwargs['f_params'])
    for key in f_params.keys():
        f_params[key] = f_params[key].detach().to(TORCH_DTYPE).requires_grad_(True)

    same_points = (ntilde == nt) and (x_given_as_xtilde or torch.equal(xtilde, x))
    eigvecs = None

    def build_kernels(th):
        C_, mask_ = localker(theta=th, theta_lower_lims=theta_lower_lims, theta_higher_lims=theta_higher_lims,
                             n_px_side=n_px_side, grad=False)
        xt_m = xtilde[:, mask_].contiguous()
        Kt = kernfun(th, xt_m, xt_m, C=C_, dC=None, diag=False)
        x_m = xt_m if same_points else x[:, mask_].contiguous()
        K_ = kernfun(th, x_m, xt_m, C=C_, dC=None, diag=False) if ntilde != nt else Kt
        Kv = kernfun(th, x_m, x2=None, C=C_, dC=None, diag=True)
        return C_, mask_, Kt, K_, Kv, x_m

    basis_route = [None]   # route of the basis in force (recorded with every tracked iteration)

    chol_factor = [None]   # (L, L^-1) of the K~ in force when its basis is the identity, else None
    estep_factor = [None]  # (K~_b, its Cholesky factor): shared by the E-steps between two kernel rebuilds
    solver_state = [None]  # the subspace solver's converged block of the previous kernel matrix (warm start)

    def project(Kt, K_):
        _BASIS.factor = None
        eigvecs_, B_, Ktb, Ktib = _stabilised_basis(Kt, start=solver_state[0])
        solver_state[0] = getattr(_BASIS, "state", None)     # the subspace solver's block, for the next EM iteration
        _BASIS.state = None
        basis_route[0] = _BASIS.route
        chol_factor[0] = _BASIS.factor if _BASIS.route == "identity" else None
        _BASIS.factor = None
        if _is_identity(B_):
            Kb = K_
            a_ = matmul(Kb, Ktib) if ntilde != nt else B_
        else:
            Kb = matmul(K_, B_)
            a_ = matmul(Kb, Ktib) if ntilde != nt else B_
        return eigvecs_, B_, Ktb, Ktib, Kb, a_

    def to_basis_vec(B_, v):
        return v.clone() if _is_identity(B_) else matmul(B_, v, transA=True)

    def to_basis_mat(B_, M):
        if _is_identity(B_):
            return M.clone()
        Mb = matmul(B_, matmul(M, B_), transA=True)
        return Mb

    rate_cache = [None]

    def lambda0_and_rate():
        """``lambda0_given_logA`` (:1874, :1934) and, from the same pass over the training points, the rate at that
        ``(logA, lambda0)`` -- what ``mean_f_given_lambda_moments`` is asked for next (:1877, :1958) with the very same
        inputs: the kernel computes both, so the second call (one launch and one wait per E-step) is answered from here."""
        f, out = _fparam_eval(lambda_m, lambda_var, r, f_params['logA'], True)
        rate_cache[0] = (lambda_m, lambda_var, _scalar(f_params['logA']), out[0], f)
        return torch.tensor(out[0], dtype=TORCH_DTYPE)

    def rate_now():
        c = rate_cache[0]
        if (c is not None and c[0] is lambda_m and c[1] is lambda_var and 'loglambda0' not in f_params
                and c[2] == _scalar(f_params['logA']) and c[3] == _scalar(f_params['lambda0'])):
            return c[4]
        return mean_f_given_lambda_moments(f_params, lambda_m, lambda_var)

    def moments_now():
        """lambda moments of the current state (utils.py:1090, 1101); with a = B = I they reduce to
        lambda_m = m, lambda_var = Kvec - diag(K~) + diag(V) and no N^3 product is formed."""
        if _is_identity(B) and ntilde == nt:
            return m_b.clone(), Kvec - torch.diagonal(K_tilde_b) + torch.diagonal(V_b)
        return lambda_moments(x_m, K_tilde_b, KKtilde_inv_b, Kvec, K_b, C, m_b, V_b, theta, kernfun=kernfun)

    if 'init_kernel' not in kwargs:
        C, mask, K_tilde, K, Kvec, x_m = build_kernels(theta)
        eigvecs, B, K_tilde_b, K_tilde_inv_b, K_b, KKtilde_inv_b = project(K_tilde, K)
    else:
        ik = kwargs['init_kernel']
        if ik.get('square') and (ntilde != nt or tuple(ik['K'].shape) != (nt, ntilde)):
            raise ValueError("varGP: this init_kernel was prepared by extend_inducing_set for a training set equal to the "
                             f"grown inducing set ({tuple(ik['K'].shape)[0]} images); got n_t = {nt}, n_tilde = {ntilde}")
        C, mask, K_tilde, K, Kvec = _cu(ik['C']), ik['mask'].to(dev), _cu(ik['K_tilde']), _cu(ik['K']), _cu(ik['Kvec'])
        B, K_tilde_b, K_b, K_tilde_inv_b = _cu(ik['B']), _cu(ik['K_tilde_b']), _cu(ik['K_b']), _cu(ik['K_tilde_inv_b'])
        # an init_kernel prepared by extend_inducing_set says which route built its basis (and brings the factor of
        # K~ when that basis is the identity); one built the notebook's way (eigh + truncation) carries neither
        basis_route[0] = ik.get('basis_route', 'eigh')
        if basis_route[0] == 'identity':
            B = _mark_identity(B)
            if ik.get('chol') is not None:
                chol_factor[0] = (_cu(ik['chol']['L']), _cu(ik['chol']['Linv']))
        KKtilde_inv_b = _cu(ik['KKtilde_inv_b']) if ntilde != nt else B
        x_m = x[:, mask].contiguous()

    m = _cu(copy.deepcopy(kwargs.get('m', torch.zeros(ntilde, dtype=TORCH_DTYPE))).detach())
    V = _cu(copy.deepcopy(kwargs.get('V', K_tilde)).detach())
    V_b = to_basis_mat(B, V) if 'V' in kwargs else K_tilde_b
    m_b = to_basis_vec(B, m)

    import os as _os
    no_fast = bool(_os.environ.get("GPFIT_NO_FAST"))

    def full_rank():
        return (not no_fast) and same_points and B.shape[0] == B.shape[1]

    lambda_m, lambda_var = moments_now()
    f_mean = mean_f_given_lambda_moments(f_params, lambda_m, lambda_var)
    loglikelihood, _, __ = compute_loglikelihood(r, f_mean, lambda_m, lambda_var, f_params)
    KL_div = compute_KL_div(m_b, V_b, K_tilde_b, K_tilde_inv_b, dK_tilde=None, ignore_warning=True)
    logmarginal = loglikelihood - KL_div

    loss_track = {k: torch.zeros(maxiter) for k in ('logmarginal', 'loglikelihood', 'KL')}
    theta_track = {key: torch.zeros(maxiter) for key in theta.keys()}
    l0key = 'lambda0' if 'lambda0' in f_params else 'loglambda0'
    f_par_track = {'logA': torch.zeros(maxiter), l0key: torch.zeros(maxiter)}
    values_track = {'loss_track': loss_track, 'theta_track': theta_track, 'f_par_track': f_par_track,
                    'variation_par_track': {'V_b': (), 'm_b': (), 'basis_route': ()}}

    def record(it):
        loss_track['loglikelihood'][it] = _scalar(loglikelihood)
        loss_track['KL'][it] = _scalar(KL_div)
        loss_track['logmarginal'][it] = _scalar(loglikelihood) - _scalar(KL_div)
        for key in theta.keys():
            theta_track[key][it] = _scalar(theta[key])
        f_par_track['logA'][it] = _scalar(f_params['logA'])
        f_par_track[l0key][it] = _scalar(f_params[l0key])
        values_track['variation_par_track']['V_b'] += (V_b.clone(),)
        values_track['variation_par_track']['m_b'] += (m_b.clone(),)
        values_track['variation_par_track']['basis_route'] += (basis_route[0],)

    times = {'estep': 0.0, 'fparams': 0.0, 'mstep': 0.0, 'kernels': 0.0, 'loss': 0.0}
    start_time_loop = time.time()
    iteration = 0
    try:
        record(0)
        print(f'Initial Loss: {-(_scalar(loglikelihood) - _scalar(KL_div)):.4f}')
        for iteration in range(1, maxiter):
            t0 = time.time()
            if nMstep > 0 and iteration > 1:
                # kernels at the theta of the last M-step, new eigenbasis, (m_b, V_b) re-projected
                C, mask, K_tilde, K, Kvec, x_m = build_kernels(theta)                       # utils.py:1803-1806
                B_old = B
                eigvecs, B, K_tilde_b, K_tilde_inv_b, K_b, KKtilde_inv_b = project(K_tilde, K)   # :1808-1818
                if _is_identity(B) and _is_identity(B_old):
                    pass                                      # B^T B_old = I: (m_b, V_b) unchanged
                else:
                    BtB = B_old if _is_identity(B) else (B.T.contiguous() if _is_identity(B_old)
                                                           else matmul(B, B_old, transA=True))
                    V_b = matmul(BtB, matmul(V_b, BtB, transB=True))                       # :1833
                    m_b = matmul(BtB, m_b)                                                 # :1840
            times['kernels'] += time.time() - t0

            t0 = time.time()
            if nEstep > 0:
                for i_estep in range(nEstep):
                    if i_estep == 0 and nMstep > 0:
                        lambda_m, lambda_var = moments_now()                                  # :1871
                        f_params['lambda0'] = lambda0_and_rate()                              # :1874
                    f_mean = rate_now()                                                         # :1877
                    fused_moments = False
                    if full_rank():
                        # fused Newton update in the original basis, then back to the eigenbasis
                        m_orig = m_b if _is_identity(B) else matmul(B, m_b)
                        m_new = torch.empty(nt, dtype=TORCH_DTYPE, device=dev)
                        V_new = torch.empty((nt, nt), dtype=TORCH_DTYPE, device=dev)
                        eng = get_engine(nt, 1)
                        rc = _lib.load().gpfit_estep(eng._ctx, _stream(), K_tilde.data_ptr(), K_tilde.stride(0), nt,
                                                     r.data_ptr(), m_orig.data_ptr(), f_mean.data_ptr(),
                                                     _scalar(f_params['logA']), m_new.data_ptr(), V_new.data_ptr(),
                                                     V_new.stride(0))
                        if rc != 0:
                            if not bool(torch.isfinite(f_mean).all()):
                                # the reference's LU solve lets NaNs through and reports them one step
                                # later, in the rate-parameter closure (utils.py:1923-1924)
                                raise ValueError(f'Nan in f_mean during f param update in Estep, closure has been '
                                                 f'called 1 times in estep {i_estep} iteration. Try substituting '
                                                 f'them with inf.')
                            raise torch.linalg.LinAlgError(f"Estep: {_lib.last_error()} (rc={rc})")
                        if _is_identity(B):
                            m_b, V_b = m_new, V_new            # symmetric by construction (gpfit_estep)
                        else:
                            m_b = matmul(B, m_new, transA=True)
                            V_b = matmul(B, matmul(V_new, B), transA=True)
                            V_b = (V_b + V_b.T) / 2
                    else:
                        # :1880, with the factor of K~_b shared by the E-steps of this iteration
                        if estep_factor[0] is None or estep_factor[0][0] is not K_tilde_b:
                            L_kb, _, _, info_kb = cholesky(K_tilde_b)
                            if info_kb != 0:
                                raise torch.linalg.LinAlgError(f"Estep: K_tilde is not positive definite (info={info_kb})")
                            estep_factor[0] = (K_tilde_b, L_kb, matmul(KKtilde_inv_b, L_kb),
                                               Kvec - torch.sum(K_b * KKtilde_inv_b, 1))
                        # the update and the moments behind it (:1884) in one device call
                        m_b, V_b, lambda_m, lambda_var = _estep_projected(r, KKtilde_inv_b, estep_factor[0][2],
                                                                          estep_factor[0][1], m_b, f_params, f_mean,
                                                                          kv0=estep_factor[0][3])
                        fused_moments = True
                    if not fused_moments:
                        lambda_m, lambda_var = moments_now()                                    # :1884
                    # (the reference evaluates the rate here, :1885; nothing reads it before the closure below overwrites it)
                    tf = time.time()
                    f_params['lambda0'] = lambda0_given_logA(f_params['logA'], r, lambda_m, lambda_var)      # :1892
                    opt_f = torch.optim.LBFGS([f_params['logA']], lr=0.1, max_iter=nFparamstep, tolerance_change=1.e-9,
                                              tolerance_grad=1.e-7, history_size=nFparamstep,
                                              line_search_fn='strong_wolfe')                                # :1897
                    calls = [0]

                    def closure_f_params():
                        calls[0] += 1
                        # one fused pass: loglik, d/dlogA with the current lambda0, and the new closed-form lambda0 (the
                        # rate vector itself is not written: nothing reads it before lambda0_and_rate() below; the
                        # gradient is assigned, so there is nothing to zero first)
                        _, out = _fparam_eval(lambda_m, lambda_var, r, f_params['logA'], False,
                                              _scalar(_lambda0_of(f_params)), want_f=False)
                        f_params['logA'].grad = torch.tensor(-out[2], dtype=TORCH_DTYPE)                     # :1913
                        f_params['lambda0'] = torch.tensor(out[6], dtype=TORCH_DTYPE)                        # :1916
                        if not math.isfinite(out[3]):
                            raise ValueError(f'Nan in f_mean during f param update in Estep, closure has been called '
                                             f'{calls[0]} times in estep {i_estep} iteration.')              # :1923
                        return torch.tensor(-out[1], dtype=TORCH_DTYPE)                                      # :1930
                    opt_f.step(closure_f_params)                                                            # :1932
                    f_params['lambda0'] = lambda0_and_rate()                                                # :1934
                    times['fparams'] += time.time() - tf
            else:
                print('No E-step')
            times['estep'] += time.time() - t0

            t0 = time.time()
            f_mean = rate_now()                                                                             # :1958
            loglikelihood, _, __ = compute_loglikelihood(r, f_mean, lambda_m, lambda_var, f_params)
            KL_div = compute_KL_div(m_b, V_b, K_tilde_b, K_tilde_inv_b, dK_tilde=None, ignore_warning=True)
            logmarginal = loglikelihood - KL_div
            times['loss'] += time.time() - t0
            record(iteration)
            print(f'Loss iter {iteration}: {-(_scalar(loglikelihood) - _scalar(KL_div)):.4f}')

            t0 = time.time()
            if nMstep > 0 and iteration < maxiter - 1:
                opt_h = torch.optim.LBFGS(theta.values(), lr=0.1, max_iter=nMstep, line_search_fn='strong_wolfe',
                                          tolerance_change=1.e-9, tolerance_grad=1.e-7, history_size=100)     # :2013
                fast = full_rank()
                if fast:
                    # (m, V) are fixed during the M-step: go to the original basis once
                    if _is_identity(B):
                        m_orig, V_orig = m_b, V_b
                    else:
                        m_orig = matmul(B, m_b)
                        V_orig = matmul(B, matmul(V_b, B, transB=True))
                        V_orig = (V_orig + V_orig.T) * 0.5
                mcalls = [0]
                v_factored = [False]

                def closure_hyperparams():
                    mcalls[0] += 1
                    opt_h.zero_grad()
                    outside = False
                    for key, value in theta.items():                                                       # :2022-2028
                        if not (theta_lower_lims[key] <= _scalar(value) <= theta_higher_lims[key]):
                            outside = True
                            print(f"{key} = {_scalar(value):.4f} is not within the limits of {theta_lower_lims[key]} and "
                                  f"{theta_higher_lims[key]}, returning inifinite loss in closure call {mcalls[0]}")
                            if theta[key].requires_grad:
                                theta[key].grad = torch.tensor(float('inf'))
                    if outside:
                        return torch.tensor(float('inf'))
                    if fast:
                        e = get_engine(nt, nx, nx)
                        res = e.fit_eval(theta, theta_lower_lims, theta_higher_lims, n_px_side, x, r, m_orig, V_orig,
                                         _scalar(f_params['logA']), _scalar(_lambda0_of(f_params)), want_grad=True,
                                         want_vectors=False, reuse_V=v_factored[0])
                        v_factored[0] = True  # V is constant for the rest of this M-step
                        loss, grad = res['loss'], res['grad']
                    elif same_points and not no_fast:
                        loss, grad = _closure_projected(theta, lims, n_px_side, x, r, B, m_b, V_b, f_params)
                    elif ntilde != nt and not no_fast:
                        loss, grad = _closure_sparse(theta, lims, n_px_side, x, xtilde, r, B, m_b, V_b, f_params)
                    else:
                        loss, grad = _closure_general(theta, lims, n_px_side, x, xtilde, r, B, m_b, V_b, f_params,
                                                      ntilde, nt)
                    for key in theta.keys():
                        if theta[key].requires_grad:
                            theta[key].grad = torch.tensor(grad[key], dtype=TORCH_DTYPE)                       # :2098-2099
                    return torch.tensor(loss, dtype=TORCH_DTYPE)

                opt_h.step(closure_hyperparams)                                                             # :2114
            elif iteration < maxiter - 1:
                print(' No M-step')
            times['mstep'] += time.time() - t0

    except (KeyboardInterrupt, Exception) as e:  # utils.py:2127-2189: roll back to the last tracked state
        print(f' ===================  Error During iteration: {iteration} =================== \n')
        fit_parameters['maxiter'] = iteration
        err_dict['is_error'] = True
        err_dict['error'] = e
        err_dict['error_message'] = repr(e)
        if iteration <= 1:
            # utils.py:2134-2138 / 2168-2172 re-raise here, but the `return` inside the reference's
            # `finally` block (utils.py:2316) swallows that exception: what the caller observes is the
            # current (not rolled-back) state with err_dict set.  Reproduced as observed.
            print('Too few iterations iterations were done to save')
        else:
            theta = {k: theta_track[k][iteration - 1].to(TORCH_DTYPE) for k in theta.keys()}
            f_params['logA'] = f_par_track['logA'][iteration - 1].to(TORCH_DTYPE)
            f_params[l0key] = f_par_track[l0key][iteration - 1].to(TORCH_DTYPE)
            V_b = values_track['variation_par_track']['V_b'][iteration - 1]
            m_b = values_track['variation_par_track']['m_b'][iteration - 1]
        # utils.py:2194-2231 (the `finally` block when is_error): kernels at the theta in force, final loss
        C, mask, K_tilde, K, Kvec, x_m = build_kernels(theta)
        eigvecs, B, K_tilde_b, K_tilde_inv_b, K_b, KKtilde_inv_b = project(K_tilde, K)
        lambda_m, lambda_var = moments_now()
        f_mean = mean_f_given_lambda_moments(f_params, lambda_m, lambda_var)
        loglikelihood = compute_loglikelihood(r, f_mean, lambda_m, lambda_var, f_params)[0]
        KL_div = compute_KL_div(m_b, V_b, K_tilde_b, K_tilde_inv_b, dK_tilde=None)
        logmarginal = loglikelihood - KL_div
        last = fit_parameters['maxiter'] - 1
        loss_track['loglikelihood'][last] = _scalar(loglikelihood)
        loss_track['KL'][last] = _scalar(KL_div)
        loss_track['logmarginal'][last] = _scalar(logmarginal)

    if fit_parameters.get('full_eigvecs', False) and (eigvecs is None or eigvecs.shape[1] != eigvecs.shape[0]):
        eigvecs = torch.linalg.eigh(K_tilde, UPLO='L')[1]     # the reference's N x N matrix, on request (utils.py:2241)
    final_kernel = {'C': C, 'mask': mask, 'K_tilde': K_tilde, 'K': K, 'Kvec': Kvec, 'eigvecs': eigvecs}
    # the factor of the final K~ (identity route only; two more n~ x n~ matrices, so by default only up to the sizes
    # of the closed-loop experiments): what extend_inducing_set grows by one row instead of refactorising
    if chol_factor[0] is not None and fit_parameters.get('keep_factor', ntilde <= 4096):
        final_kernel['chol'] = {'L': chol_factor[0][0], 'Linv': chol_factor[0][1]}
    if not is_simmetric(V_b, 'V_b'):
        print('Final V_b is not simmetric, maximum difference: ', torch.max(torch.abs(V_b - V_b.T)))
        V_b = (V_b + V_b.T) / 2
    if not is_posdef(V_b, 'V_b'):
        print('Final V_b is not posdef, this should not be possible if you are skipping the last M-step')
        V_b = V_b + torch.eye(V_b.shape[0], dtype=TORCH_DTYPE, device=V_b.device) * EIGVAL_TOL

    print(f'\nTime spent for E-steps:       {times["estep"]:.3f}s,')
    print(f'Time spent for f params:      {times["fparams"]:.3f}s')
    print(f'Time spent for m / V update:  {times["estep"] - times["fparams"]:.3f}s')
    print(f'Time spent for M-steps:       {times["mstep"]:.3f}s')
    print(f'Time spent for All-steps:     {times["estep"] + times["mstep"]:.3f}s')
    print(f'Time spent computing Kernels: {times["kernels"]:.3f}s')
    print(f'Time spent computing Loss:    {times["loss"]:.3f}s')
    print(f'\nTime total after init:        {time.time() - start_time_loop:.3f}s')
    print(f'Time total before init:       {time.time() - start_time_before_init:.3f}s')
    print(f'Final Loss: {-_scalar(logmarginal):.4f}')

    nkeep = fit_parameters['maxiter']
    for key in values_track.keys():
        for sub in values_track[key].keys():
            values_track[key][sub] = values_track[key][sub][:nkeep]

    fit_model = {
        'fit_parameters': fit_parameters, 'final_kernel': final_kernel, 'err_dict': err_dict, 'xtilde': xtilde,
        'hyperparams_tuple': (theta, theta_lower_lims, theta_higher_lims), 'f_params': f_params, 'm_b': m_b,
        'V_b': V_b, 'C': C, 'mask': mask, 'K_tilde_b': K_tilde_b, 'K_tilde_inv_b': K_tilde_inv_b, 'K_b': K_b,
        'Kvec': Kvec, 'B': B, 'values_track': values_track, 'basis_route': basis_route[0],
    }
    return fit_model, err_dict


@torch.no_grad()
def test(X_test, R_test, xtilde, X_train=None, at_iteration=None, **kwargs):
    """Predict the firing rate of one cell on test images from a ``fit_model`` (utils.py:326-412):
    ``test(X_test, R_test, X_train=X, at_iteration=None, **fit_model)``.  X_test is
    [n_images, n_px, n_px, 1]; all images go through ``lambda_moments_star`` in one batch."""
    fp = kwargs['fit_parameters']
    maxiter, nEstep, nMstep = fp.get('maxiter', 0), fp.get('nEstep', 0), fp.get('nMstep', 0)
    cellid, n_px_side = fp.get('cellid'), fp.get('n_px_side')
    mask, C = kwargs.get('mask'), kwargs.get('C')
    theta, theta_lower_lims, theta_higher_lims = kwargs.get('hyperparams_tuple')
    m, V, B = kwargs.get('m_b'), kwargs.get('V_b'), kwargs.get('B')
    K_tilde, K_tilde_inv = kwargs.get('K_tilde_b'), kwargs.get('K_tilde_inv_b')
    f_params = kwargs.get('f_params')
    xtilde = _cu(xtilde)
    reserve(xtilde.shape[0], xtilde.shape[1], xtilde.shape[1])
    A = math.exp(_scalar(f_params['logA']))
    lambda0 = _scalar(_lambda0_of(f_params))

    if at_iteration is not None and X_train is not None:                                   # utils.py:358-386
        vt = kwargs['values_track']
        theta = {key: val[at_iteration] for key, val in vt['theta_track'].items()}
        m = vt['variation_par_track']['m_b'][at_iteration]
        V = vt['variation_par_track']['V_b'][at_iteration]
        A = math.exp(_scalar(vt['f_par_track']['logA'][at_iteration]))
        lambda0 = (_scalar(vt['f_par_track']['lambda0'][at_iteration]) if 'lambda0' in vt['f_par_track']
                   else math.exp(_scalar(vt['f_par_track']['loglambda0'][at_iteration])))
        C, mask = localker(theta=theta, theta_higher_lims=theta_higher_lims, theta_lower_lims=theta_lower_lims,
                           n_px_side=n_px_side, grad=False)
        xt_m = xtilde[:, mask].contiguous()
        Kt = acosker(theta, xt_m, xt_m, C=C, diag=False)
        routes = vt['variation_par_track'].get('basis_route')     # absent in a dict the reference itself produced
        _, B, K_tilde, K_tilde_inv = _stabilised_basis(Kt, route=routes[at_iteration] if routes else None)

    X_test = _cu(X_test)
    n_img = X_test.shape[0]
    xstar = X_test.reshape(n_img, -1)                                                      # utils.py:389-390
    mask = mask.to(xstar.device)
    mu_star, sigma_star2 = lambda_moments_star(xstar[:, mask].contiguous(), xtilde[:, mask].contiguous(), C, theta,
                                               K_tilde, K_tilde_inv, m, V, B, 'acosker')   # :393
    R_predicted = torch.exp(A * mu_star + 0.5 * A * A * sigma_star2 + lambda0)              # :395
    R_test = _cu(R_test)
    r2, sigma_r2 = explained_variance(R_test[:, :, cellid], R_predicted, sigma=True)        # :400
    print(f"\n\n Pietro's model: R2 = {float(r2):.2f} ± {float(sigma_r2):.2f} Cell: {cellid} maxiter = {maxiter}, "
          f"nEstep = {nEstep}, nMstep = {nMstep} \n")
    return R_test[:, :, cellid], R_predicted, r2, sigma_r2
