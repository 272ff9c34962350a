"""ctypes binding of libgpfit_mi355x.so (the C ABI declared in include/gpfit_mi355x.h).

The product path has NO fallback: if the HIP library is missing or a call fails, an
exception is raised."""
from __future__ import annotations

import ctypes
import os

from .build import lib_path

vp, i32, i64, f64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double
pd = ctypes.POINTER(ctypes.c_double)

_SIGS = {
    "gpfit_version": (i32, []),
    "gpfit_last_error": (ctypes.c_char_p, []),
    "gpfit_dgemm": (i32, [vp, i32, i32, i32, i32, i32, f64, vp, i64, vp, i64, f64, vp, i64, i32, i32, i32]),
    "gpfit_dgemm_ex": (i32, [vp, i32, i32, i32, i32, i32, f64, vp, i64, vp, i64, f64, vp, i64, i32, i32, i32, i32, i32]),
    "gpfit_ctx_create": (i32, [i32, i64, i64, i64, ctypes.POINTER(vp)]),
    "gpfit_ctx_destroy": (None, [vp]),
    "gpfit_check_limits": (i32, [pd, pd, pd]),
    "gpfit_localker_mask": (i32, [pd, i32, i32, vp, ctypes.POINTER(i64)]),
    "gpfit_localker": (i32, [vp, vp, pd, i32, i32, vp, i64, vp, vp]),
    "gpfit_acosker": (i32, [vp, vp, f64, vp, i64, i64, vp, i64, i64, i64, vp, i64, vp, vp, i64, vp]),
    "gpfit_acosker_pullback": (i32, [vp, vp, f64, vp, i64, i64, vp, i64, i64, i64, vp, i64, vp, i64, vp, vp, i64, pd]),
    "gpfit_acosker_diag": (i32, [vp, vp, f64, vp, i64, i64, i64, vp, i64, vp, vp, vp]),
    "gpfit_fit_eval": (i32, [vp, vp, pd, pd, pd, i32, i32, vp, i64, i64, vp, vp, vp, i64, f64, f64, i32, pd,
                             vp, vp, vp]),
    "gpfit_potrf": (i32, [vp, vp, vp, i64, i64, vp, i64, vp, i64, pd, ctypes.POINTER(i32)]),
    "gpfit_potrf_append": (i32, [vp, vp, vp, i64, vp, i64, i64, vp, pd, ctypes.POINTER(i32)]),
    "gpfit_estep": (i32, [vp, vp, vp, i64, i64, vp, vp, vp, f64, vp, vp, i64]),
    "gpfit_estep_projected": (i32, [vp, vp, vp, i64, vp, i64, vp, i64, i64, i64, vp, vp, vp, f64, vp, vp, i64, vp, vp,
                                    vp]),
    "gpfit_fparam_eval": (i32, [vp, vp, vp, vp, vp, i64, f64, i32, f64, vp, pd]),
    "gpfit_set_profile": (i32, [vp, i32]),
    "gpfit_get_profile": (i32, [vp, pd]),
    "gpfit_get_phases": (i32, [vp, pd]),
    "gpfit_last_enqueue_ms": (f64, [vp]),
    "gpfit_fit_eval_f32": (i32, [vp, vp, pd, pd, pd, i32, i32, vp, i64, i64, vp, vp, vp, i64, f64, f64, i32, pd,
                                 vp, vp, vp]),
    "gpfit_fit_eval_finish": (i32, [vp, pd]),
    "gpfit_fit_eval_batch": (i32, [vp, i32, vp, pd, pd, pd, i32, i32, vp, i64, i64, vp, vp, vp, i64, pd, pd, i32, pd,
                                   ctypes.POINTER(i32)]),
    "gpfit_fit_eval_batch_f32": (i32, [vp, i32, vp, pd, pd, pd, i32, i32, vp, i64, i64, vp, vp, vp, i64, pd, pd, i32, pd,
                                       ctypes.POINTER(i32)]),
    "gpfit_fit_eval_projected": (i32, [vp, vp, pd, pd, pd, i32, i32, vp, i64, i64, vp, vp, i64, i64, vp, vp, i64, f64, f64, pd]),
    "gpfit_fit_eval_sparse": (i32, [vp, vp, pd, pd, pd, i32, i32, vp, i64, i64, vp, i64, i64, vp, vp, i64, i64, vp, vp, i64,
                                    f64, f64, pd]),
    "gpfit_grad_pullback": (i32, [vp, vp, pd, i32, i32, vp, i64, i64, vp, i64, vp, pd]),
    "gpfit_nd_utility": (i32, [vp, vp, vp, i64, vp, i32, vp]),
    "gpfit_probe_mfma_f64": (i32, [vp, vp, i32, i32]),
    "gpfit_probe_stream_copy": (i32, [vp, vp, vp, i64]),
}

_lib = None


class GpfitError(RuntimeError):
    pass


def exported_symbols():
    return sorted(_SIGS)


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise GpfitError(
            f"{path} not found: build it with `python -m gaussian_processes_amd.build` "
            "(there is no CPU fallback for the GP fit path)")
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().gpfit_last_error().decode(errors="replace")


def check(rc: int, what: str):
    if rc < 0 and rc != -2:
        raise GpfitError(f"{what} failed (rc={rc}): {last_error()}")
    return rc


def darr(values):
    return (ctypes.c_double * len(values))(*[float(v) for v in values])
