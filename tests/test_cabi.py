"""CPU-side checks of the drop-in boundary: the shared library builds for gfx950, loads, and
exports every symbol include/gpfit_mi355x.h declares; host-only entry points behave like the
reference.  No compute call touches a GPU here."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden
from gaussian_processes_amd import _lib, synthetic as syn
from gaussian_processes_amd.build import build_library

KEYS = syn.THETA_KEYS


@pytest.fixture(scope="module")
def lib():
    build_library(verbose=False)
    return _lib.load()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "gpfit_mi355x.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(gpfit_\w+)\s*\(", hdr)))
    assert len(declared) >= 8
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert set(declared) == set(_lib.exported_symbols()), "ctypes table and header disagree"
    assert lib.gpfit_version() == 100


def test_check_limits_matches_reference_rule(lib):
    lo, up = syn.limits()
    th = syn.theta_eval()
    v = lambda d_: _lib.darr([d_[k] for k in KEYS])
    assert lib.gpfit_check_limits(v(th), v(lo), v(up)) == 0
    th["eps_0y"] = -1.0  # boundary is inside (utils.py:866 uses <=)
    assert lib.gpfit_check_limits(v(th), v(lo), v(up)) == 0
    th["eps_0y"] = -1.0000001
    assert lib.gpfit_check_limits(v(th), v(lo), v(up)) == -2
    assert b"eps_0y" in lib.gpfit_last_error()
    th = syn.theta_eval()
    th["Amp"] = float("nan")
    assert lib.gpfit_check_limits(v(th), v(lo), v(up)) == -2


def test_localker_mask_matches_golden(lib):
    g = load_golden("g1_localker.npz")
    for i in range(int(g["n_cases"])):
        n_px = int(g[f"c{i}_n_px"])
        buf = (ctypes.c_uint8 * (n_px * n_px))()
        d = ctypes.c_int64()
        rc = lib.gpfit_localker_mask(_lib.darr(g[f"c{i}_theta"]), n_px, n_px, buf, ctypes.byref(d))
        assert rc == 0
        mask = np.array(list(buf), dtype=bool)
        assert np.array_equal(mask, g[f"c{i}_mask"])
        assert int(d.value) == int(g[f"c{i}_mask"].sum())


def test_missing_gpu_fails_loudly():
    import torch
    from gaussian_processes_amd.engine import GPFitEngine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.GpfitError):
        GPFitEngine(64, 16)


def test_dropin_module_imports_and_refuses_to_run_without_gpu():
    """The host module must import anywhere (so build() can check it) but every compute entry
    point fails loudly without a GPU -- there is no CPU fallback on the product path."""
    import torch
    from gaussian_processes_amd import utils as gp
    for name in ("localker", "acosker", "lambda_moments", "mean_f_given_lambda_moments", "mean_f", "lambda0_given_logA",
                 "compute_loglikelihood", "log_det", "compute_KL_div", "Estep", "lambda_moments_star", "varGP", "test",
                 "generate_theta", "generate_xtilde", "explained_variance", "is_posdef", "is_simmetric", "safe_log"):
        assert callable(getattr(gp, name)), name
    assert gp.EIGVAL_TOL == 1e-4 and gp.MIN_TOLERANCE == 1e-11 and gp.TORCH_DTYPE == torch.float64
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lo, up = syn.limits()
    with pytest.raises(_lib.GpfitError):
        gp.localker(syn.theta0(), up, lo, 8)


def test_product_package_never_imports_the_oracle():
    """oracle/ is test infrastructure: no file of the product package may import it."""
    import glob
    pkg = os.path.join(ROOT, "gaussian_processes_amd")
    for f in glob.glob(os.path.join(pkg, "**", "*.py"), recursive=True):
        src = open(f).read()
        assert "import oracle" not in src and "from oracle" not in src, f
