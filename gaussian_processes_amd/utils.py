"""Drop-in host module for the GP fit path of ``Spatial_GP_repo/utils.py``.

Same function names, argument meaning and error behaviour as the reference for the path
SURVEY.md section 8 scopes (``localker``, ``acosker``, ``lambda_moments``, ``compute_KL_div``,
``Estep``, ``varGP``, ``test`` ...), but every kernel / factorisation / solve runs in the
hand-written HIP library ``libgpfit_mi355x.so`` through ctypes.  torch supplies device
memory, streams and (for the rank decision only) ``torch.linalg.eigh``.

There is no CPU fallback: without a GPU or without the built library every entry point raises.
Out of scope (SURVEY.md section 2 rows 10-14,16-19): plotting, pickling helpers, the dataset
container, the active-learning utility.
"""
from __future__ import annotations

import copy
import ctypes
import math
import warnings

import torch

from . import _lib
from .engine import GPFitEngine, _grid, theta_vec
from .synthetic import THETA_KEYS

torch.set_grad_enabled(False)  # reference utils.py:2 (analytic gradients only)

TORCH_DTYPE = torch.float64          # utils.py:31
MIN_TOLERANCE = 1.e-11               # utils.py:37
EIGVAL_TOL = 1.e-4                   # utils.py:39 (module global read at call time, as in the reference)
PI32 = 3.1415927410125732            # utils.py:25: float32-rounded pi

# dC dict key order of the reference (utils.py:910) == matrix order of gpfit_localker
DC_KEYS = ("Amp", "-2log2beta", "-log2rho2", "eps_0x", "eps_0y")


def _device():
    if not torch.cuda.is_available():
        raise _lib.GpfitError("gaussian_processes_amd.utils needs an MI355X: there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


class _EnginePool:
    """One growing workspace per device (the C context is not re-entrant: one per thread)."""

    def __init__(self):
        self.eng = {}

    def get(self, n, d, d_full=None):
        dev = _device()
        key = dev.index
        d_full = int(d_full or d)
        e = self.eng.get(key)
        if e is None or e.n_max < n or e.d_max < d or e.d_full_max < d_full:
            n_cap = max(n, e.n_max if e else 0)
            d_cap = max(d, e.d_max if e else 0)
            f_cap = max(d_full, e.d_full_max if e else 0)
            if e is not None:
                e.close()
            e = GPFitEngine(n_cap, d_cap, f_cap, device=key)
            self.eng[key] = e
        return e


_POOL = _EnginePool()


def get_engine(n, d, d_full=None) -> GPFitEngine:
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
