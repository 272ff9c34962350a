"""CPU oracle for the GP fit hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a CPU restatement (torch CPU fp64, same BLAS/LAPACK the reference
itself runs on) of the arithmetic in the reference's ``Spatial_GP_repo/utils.py``
hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; the product package
(``gaussian_processes_amd``) never does.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real
reference in the build container, runs it on seeded synthetic inputs and
commits the inputs/outputs as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function below against those vectors, and against the known
answers saved in the reference's ``moments_gradients.ipynb`` (cells 1-3).

Two formulations of the unit of work (one M-step closure evaluation,
reference ``utils.py:2017-2112``) are given:

* ``mstep_closure_reference``  -- the reference's own operation sequence
  (materialised dK{6}, projection on the eigenbasis B, LU inverse, 13+13 GEMM
  gradient products).  Used for the fixtures, for the truncated-rank mode and
  as the ``cpu_baseline`` that ``bench.py`` times.
* ``mstep_closure_cholesky``   -- the original-basis Cholesky formulation the
  HIP path implements (full-rank, n_tilde == n_t).  Algebraically identical
  when B is orthogonal; used as the parity checker at sizes where the
  reference formulation would take minutes.

Every function cites the reference lines it restates.
"""
from __future__ import annotations

import math
import warnings

import torch

F64 = torch.float64

# utils.py:25 overwrites torch.pi *before* the default dtype becomes float64
# (utils.py:33), so every use of pi in the reference is the float32-rounded value.
PI32 = 3.1415927410125732

THETA_KEYS = ("sigma_0", "eps_0x", "eps_0y", "-2log2beta", "-log2rho2", "Amp")
# localker returns dC for these five (utils.py:910); sigma_0 is handled in acosker.
DC_KEYS = ("Amp", "-2log2beta", "-log2rho2", "eps_0x", "eps_0y")

EIGVAL_TOL_DEFAULT = 1.0e-4  # utils.py:39
MASK_THRESHOLD = 0.001  # utils.py:883


def _f(v) -> float:
    return float(v.item()) if hasattr(v, "item") else float(v)


def _t(a) -> torch.Tensor:
    return torch.as_tensor(a, dtype=F64)


# --------------------------------------------------------------------------- a1
def linspace_pm1(n: int) -> torch.Tensor:
    """torch.linspace(-1, 1, n) in fp64 as ATen evaluates it -- two-sided
    (``fma(step, i, start)`` below the midpoint, ``fma(-step, n-1-i, end)`` above) --
    so that the pixel mesh of utils.py:876 is reproduced to the last bit.  The fused
    multiply-add is emulated through x87 extended precision (Python 3.10 has no
    math.fma); verified equal to torch.linspace in tests/test_oracle_golden.py."""
    import numpy as np
    if n == 1:
        return torch.tensor([-1.0], dtype=F64)
    step = np.float64(2.0) / np.float64(n - 1)
    L = np.longdouble
    half = n // 2
    vals = [float(np.float64(L(-1.0) + L(step) * L(i))) if i < half
            else float(np.float64(L(1.0) - L(step) * L(n - 1 - i))) for i in range(n)]
    return torch.tensor(vals, dtype=F64)


def pixel_grid(n_px_side):
    """Flattened (xcord, ycord) of the 'ij' mesh (utils.py:876-879): y follows the
    row index, x the column index.  ``n_px_side`` may be an int (square image, as
    in the reference) or ``(n_rows, n_cols)`` -- the rectangular generalisation
    SURVEY 7.3(5) asks for (d=128 -> 16x8)."""
    if isinstance(n_px_side, (tuple, list)):
        n_rows, n_cols = int(n_px_side[0]), int(n_px_side[1])
    else:
        n_rows = n_cols = int(n_px_side)
    ys = linspace_pm1(n_rows)
    xs = linspace_pm1(n_cols)
    ycord = ys[:, None].expand(n_rows, n_cols).reshape(-1).clone()
    xcord = xs[None, :].expand(n_rows, n_cols).reshape(-1).clone()
    return xcord, ycord


def check_limits(theta, lower, upper):
    """utils.py:865-867 -- ValueError when any hyperparameter leaves its box."""
    for k, v in theta.items():
        if not (lower[k] <= _f(v) <= upper[k]):
            raise ValueError(
                f"{k} = {_f(v):.4f} is not within the limits of {lower[k]} and {upper[k]}")


def spatial_metric(theta, lower, upper, n_px_side, grad=False):
    """``localker`` (utils.py:861-914): the squared-exponential spatial metric
    C[d,d] over pixel coordinates, the pixel mask and (optionally) dC/dtheta."""
    check_limits(theta, lower, upper)
    ex, ey = _f(theta["eps_0x"]), _f(theta["eps_0y"])
    eb = math.exp(_f(theta["-2log2beta"]))
    er = math.exp(_f(theta["-log2rho2"]))
    amp = _f(theta["Amp"])
    xc, yc = pixel_grid(n_px_side)

    log_alpha = -eb * ((xc - ex) ** 2 + (yc - ey) ** 2)  # :880
    alpha = torch.exp(log_alpha)  # :881
    mask = alpha >= MASK_THRESHOLD  # :883
    alpha, log_alpha, xc, yc = alpha[mask], log_alpha[mask], xc[mask], yc[mask]

    log_smooth = -er * ((xc[None, :] - xc[:, None]) ** 2 + (yc[None, :] - yc[:, None]) ** 2)  # :890
    C = amp * alpha[:, None] * torch.exp(log_smooth) * alpha[None, :]  # :892-895
    C = (C + C.T) / 2  # :898
    if not grad:
        return C, mask
    dC = {
        "Amp": C / amp,  # :902
        "eps_0x": 2.0 * eb * C * (xc[:, None] + xc[None, :] - 2 * ex),  # :904
        "eps_0y": 2.0 * eb * C * (yc[:, None] + yc[None, :] - 2 * ey),  # :905
        "-2log2beta": C * (log_alpha[:, None] + log_alpha[None, :]),  # :907
        "-log2rho2": C * log_smooth,  # :909
    }
    return C, mask, {k: dC[k] for k in DC_KEYS}


# --------------------------------------------------------------------------- a2
def arccos_gram(theta, x1, x2, C, dC=None):
    """``acosker`` with diag=False (utils.py:968-1025).  x1[n1,d], x2[n2,d] are
    already masked.  Returns K[n1,n2] (symmetrised iff n1 == n2) and, when dC is
    given, the six un-symmetrised derivative matrices."""
    s0 = _f(theta["sigma_0"])
    x1, x2 = _t(x1), _t(x2)
    x1C = x1 @ C
    q1 = torch.sqrt((x1C * x1).sum(1) + s0 * s0)  # :978
    q2 = torch.sqrt(((x2 @ C) * x2).sum(1) + s0 * s0)  # :979
    qq = torch.outer(q1, q2)  # :981
    G = x1C @ x2.T + s0 * s0  # :982
    cosd = torch.clip(G / (qq + 1e-7), -1, 1)  # :984
    delta = torch.arccos(cosd)  # :986
    J = (torch.sqrt(1 - cosd * cosd) + PI32 * cosd - delta * cosd) / PI32  # :988
    K = qq * J  # :990
    dK = None
    if dC is not None:
        dK = {}
        dqq = s0 * s0 * (q2[None, :] / q1[:, None] + q1[:, None] / q2[None, :])  # :996
        dcos = (2 * s0 * s0 - cosd * dqq) / qq  # :998
        dJ = -(delta - PI32) * dcos / PI32  # :1000
        dK["sigma_0"] = (qq * dJ + dqq * J) / s0  # :1004
        for key, dCk in dC.items():
            if key == "sigma_0":
                continue
            x1D = x1 @ dCk
            dq1 = 0.5 * (x1D * x1).sum(1) / q1  # :1012
            dq2 = 0.5 * ((x2 @ dCk) * x2).sum(1) / q2  # :1013
            dqq = dq1[:, None] * q2[None, :] + q1[:, None] * dq2[None, :]  # :1015
            dcos = (x1D @ x2.T - cosd * dqq) / qq  # :1017
            dJ = -(delta - PI32) * dcos / PI32  # :1019
            dK[key] = qq * dJ + dqq * J  # :1021
    if x1.shape[0] == x2.shape[0]:
        K = (K + K.T) / 2  # :1024-1025 (dK is NOT symmetrised)
    return (K, dK) if dC is not None else K


# --------------------------------------------------------------------------- a3
def arccos_gram_diag(theta, x1, C, dC=None):
    """``acosker`` with diag=True (utils.py:1027-1044): Kvec_i = x_i C x_i + s0^2."""
    s0 = _f(theta["sigma_0"])
    x1 = _t(x1)
    Kvec = ((x1 @ C) * x1).sum(1) + s0 * s0  # :1029
    if dC is None:
        return Kvec
    dKvec = {"sigma_0": torch.full_like(Kvec, 2 * s0 * s0 / s0)}  # :1036
    for key, dCk in dC.items():
        if key == "sigma_0":
            continue
        dKvec[key] = ((x1 @ dCk) * x1).sum(1)  # :1042
    return Kvec, dKvec


# --------------------------------------------------------------------------- a5
def latent_moments(a, K, Kvec, m, V, dK=None, dKt=None, dKvec=None, Kt_inv=None):
    """``lambda_moments`` (utils.py:1072-1124): lam_m = a m,
    lam_var = Kvec - diag(K a^T) + diag(a V a^T); and their theta-gradients."""
    lam_m = a @ m  # :1090
    aV = a @ V
    lam_var = Kvec + (-(K * a) + a * aV).sum(1)  # :1101 (V symmetric => a V == (V a^T)^T)
    if dK is None or dKt is None or dKvec is None or Kt_inv is None:
        return lam_m, lam_var
    dlam_m, dlam_var = {}, {}
    Va = V @ a.T
    for key in dK:
        da = (dK[key] - a @ dKt[key]) @ Kt_inv  # :1114
        dlam_m[key] = da @ m  # :1117
        dlam_var[key] = (dKvec[key] + 2 * (da * Va.T).sum(1)
                         - (dK[key] * a).sum(1) - (K * da).sum(1))  # :1120
    return lam_m, lam_var, dlam_m, dlam_var


# --------------------------------------------------------------------------- a6
def rate_mean(logA, lambda0, lam_m, lam_var):
    """``mean_f_given_lambda_moments`` (utils.py:1126-1141)."""
    A = math.exp(_f(logA))
    return torch.exp(A * lam_m + 0.5 * A * A * lam_var + _f(lambda0))


def lambda0_closed_form(logA, r, lam_m, lam_var):
    """``lambda0_given_logA`` (utils.py:1215-1229)."""
    A = math.exp(_f(logA))
    return float(torch.log(r.sum()) - torch.log(torch.exp(A * lam_m + 0.5 * A * A * lam_var).sum()))


# --------------------------------------------------------------------------- a7
def expected_loglik(r, f, lam_m, lam_var, logA, lambda0, dlam_m=None, dlam_var=None,
                    f_param_grad=False):
    """``compute_loglikelihood`` (utils.py:1231-1269): L = A r.lam_m + lambda0 sum(r) - sum(f)."""
    A = math.exp(_f(logA))
    l0 = _f(lambda0)
    r_lm = r @ lam_m
    sum_r = r.sum()
    L = A * r_lm + l0 * sum_r - f.sum()  # :1243
    if f_param_grad:
        g = {"logA": A * (r_lm - torch.dot(lam_m + A * lam_var, f)),  # :1253
             "lambda0": sum_r - f.sum()}  # :1255
        return L, g
    if dlam_m is not None and dlam_var is not None:
        dL = {k: A * (r @ dlam_m[k]) - A * (f @ dlam_m[k]) - 0.5 * A * A * (f @ dlam_var[k])
              for k in dlam_m}  # :1266
        return L, dL
    return L, r_lm, sum_r


# --------------------------------------------------------------------------- a8
def guarded_log(x):
    """``safe_log`` (utils.py:665-673)."""
    if torch.any(x <= 0):
        raise ValueError("Negative or zero input to log detected")
    if torch.any(x < 1e-10):
        raise ValueError("Very small input to log detected")
    return torch.log(x)


def chol_logdet(M, tol=EIGVAL_TOL_DEFAULT, quiet=False):
    """``log_det`` (utils.py:1271-1304): 2*sum(log diag chol(M)); on a failed
    factorisation fall back to the log of the eigenvalues above the truncation
    rule (symmetric M) or 0 with a warning (non-symmetric M)."""
    try:
        U = torch.linalg.cholesky(M, upper=True)  # :1275
        return 2 * guarded_log(torch.diagonal(U)).sum()  # :1278
    except Exception:
        if bool(torch.all((M - M.T).abs() <= 1e-11)):  # is_simmetric, :657-663
            ev = torch.linalg.eigvalsh(M)
            keep = ev > max(float(ev.max()) * tol, tol)  # :1287
            if not quiet:
                warnings.warn("matrix in logdet is symmetric but not posdef, using eigendecomposition")
            return guarded_log(ev[keep]).sum()  # :1301
        warnings.warn("matrix in logdet is not symmetric")
        return torch.tensor(0.0, dtype=F64)  # :1304


# --------------------------------------------------------------------------- a9
def kl_divergence(m, V, Kt, Kt_inv, dKt=None, tol=EIGVAL_TOL_DEFAULT, quiet=False):
    """``compute_KL_div`` (utils.py:1306-1337).  No -n/2 term (utils.py:1326)."""
    c = V @ Kt_inv  # :1318
    b = Kt_inv @ m  # :1320
    KL = (-0.5 * chol_logdet(V, tol, quiet) + 0.5 * chol_logdet(Kt, tol)
          + 0.5 * (m @ b) + 0.5 * torch.trace(c))  # :1326
    if dKt is None:
        return KL
    dKL = {}
    for key, dk in dKt.items():
        Bk = dk @ Kt_inv  # :1331
        dKL[key] = 0.5 * torch.trace(Bk) - 0.5 * torch.trace(c @ Bk) - 0.5 * (b @ (Bk @ m))  # :1333
    return KL, dKL


# --------------------------------------------------------------------------- a10
def newton_estep(r, a, m, logA, f, Kt):
    """``Estep`` alpha=1 / update_V_inv=False branch (utils.py:1420-1439)."""
    A = math.exp(_f(logA))
    g = A * (a.T @ (r - f))  # :1421
    G = A * A * (a.T @ (a * f[:, None]))  # :1422
    n = Kt.shape[0]
    V_new = torch.linalg.solve(torch.eye(n, dtype=F64) + Kt @ G, Kt)  # :1430
    m_new = V_new @ (G @ m + g)  # :1431
    V_new = (V_new + V_new.T) / 2  # :1438
    return m_new, V_new


# --------------------------------------------------------------------------- a11
def predict_moments(theta, xstar, xtilde, C, Kt, Kt_inv, m, V, B):
    """``lambda_moments_star`` (utils.py:1476-1500) for a batch of rows xstar[n*,d]
    (the reference loops over single rows, utils.py:388-397; row results are
    independent so batching is exact)."""
    ks = arccos_gram(theta, xstar, xtilde, C) @ B  # :1486-1487
    a = ks @ Kt_inv  # :1489
    mu = a @ m  # :1491
    kss = arccos_gram_diag(theta, xstar, C)  # :1494
    s2 = kss + ((a @ (V - Kt)) * a).sum(1)  # :1498
    return mu, s2


def predict_rate(logA, lambda0, mu, s2):
    """utils.py:395."""
    A = math.exp(_f(logA))
    return torch.exp(A * mu + 0.5 * A * A * s2 + _f(lambda0))


# --------------------------------------------------------------------------- a4
def eigen_basis(Kt, tol=EIGVAL_TOL_DEFAULT):
    """Spectral stabilisation (utils.py:1682-1694): eigh, keep lambda > max(lmax*tol, tol)."""
    ev, evec = torch.linalg.eigh(Kt, UPLO="L")  # :1682
    keep = ev > max(float(ev.max()) * tol, tol)  # :1683 (strict >)
    return ev, evec, keep


# --------------------------------------------------------------------------- a12
def mstep_closure_reference(theta, lower, upper, n_px_side, x, xtilde, r, B, m_b, V_b,
                            logA, lambda0, tol=EIGVAL_TOL_DEFAULT, want_parts=False):
    """One evaluation of ``closure_hyperparams`` (utils.py:2017-2112), in the
    reference's own formulation.  x[nt,nx_full], xtilde[ntilde,nx_full] are
    UN-masked; B[ntilde,n] is the (fixed) eigenbasis.  Returns
    ``(loss, grad)`` with loss = -(loglik - KL) and grad[k] = d loss / d theta_k;
    out-of-box theta returns (inf, inf...) as utils.py:2020-2028 does."""
    for k, v in theta.items():
        if not (lower[k] <= _f(v) <= upper[k]):
            inf = float("inf")
            return inf, {kk: inf for kk in theta}
    nt, ntilde = x.shape[0], xtilde.shape[0]
    C, mask, dC = spatial_metric(theta, lower, upper, n_px_side, grad=True)  # :2030
    xm, xtm = _t(x)[:, mask], _t(xtilde)[:, mask]
    Kt, dKt = arccos_gram(theta, xtm, xtm, C, dC)  # :2031
    if ntilde != nt:
        K, dK = arccos_gram(theta, xm, xtm, C, dC)  # :2032
    else:
        K, dK = Kt, dKt
    Kvec, dKvec = arccos_gram_diag(theta, xm, C, dC)  # :2033

    Kt_b = B.T @ Kt @ B  # :2047
    Kt_b = (Kt_b + Kt_b.T) * 0.5  # :2048
    K_b = K @ B  # :2049
    dKt_b = {k: B.T @ dKt[k] @ B for k in dKt}  # :2061
    dK_b = {k: dK[k] @ B for k in dK}  # :2062
    Kt_inv_b = torch.linalg.solve(Kt_b, torch.eye(Kt_b.shape[0], dtype=F64))  # :2067
    a = K_b @ Kt_inv_b if ntilde != nt else B  # :2068

    lam_m, lam_var, dlam_m, dlam_var = latent_moments(
        a, K_b, Kvec, m_b, V_b, dK_b, dKt_b, dKvec, Kt_inv_b)  # :2070 -> 1180
    f = rate_mean(logA, lambda0, lam_m, lam_var)
    L, dL = expected_loglik(r, f, lam_m, lam_var, logA, lambda0, dlam_m, dlam_var)  # :2085
    KL, dKL = kl_divergence(m_b, V_b, Kt_b, Kt_inv_b, dKt_b, tol)  # :2086
    loss = -(L - KL)  # :2087-2089
    grad = {k: -(dL[k] - dKL[k]) for k in theta}  # :2097-2099
    if want_parts:
        return float(loss), {k: float(grad[k]) for k in THETA_KEYS}, dict(
            loglik=float(L), KL=float(KL), lam_m=lam_m, lam_var=lam_var, f=f,
            Kt=Kt, Kvec=Kvec, C=C, mask=mask)
    return float(loss), {k: float(grad[k]) for k in THETA_KEYS}


# ------------------------------------------------------------ Cholesky restatement
def mstep_closure_cholesky(theta, lower, upper, n_px_side, x, r, m, V, logA, lambda0,
                           want_grad=True, want_parts=False):
    """The same unit of work in the ORIGINAL basis, full rank, n_tilde == n_t
    (SURVEY 7.2).  With K~ = L L^T, b = K~^-1 m, W = 1/2 (K~^-1 (K~-V) K~^-1 - b b^T):

      lam_m  = m,   lam_var = Kvec - diag(K~) + diag(V)
      KL     = -1/2 log|V| + 1/2 log|K~| + 1/2 m.b + 1/2 tr(K~^-1 V)
      dKL_p  = sum_ij W_ij dK~_p,ij
      dL_p   = -1/2 A^2 sum_i f_i (dKvec_p,i - dK~_p,ii)

    and the contraction with dK~_p is pulled back to the d x d metric:
      sum_ij W_ij dK~_p,ij = <dC_p, X^T (A_w + diag(t)) X>            (p != sigma_0)
    with A_w = W o (pi-delta)/pi, u_i = sum_j (W o sqrt(1-c^2)/pi)_ij q_j, t_i = u_i/q_i
    (derived from utils.py:1012-1021; the sigma_0 row from utils.py:996-1004).
    m, V here are in the original basis (m = B m_b, V = B V_b B^T).
    """
    for k, v in theta.items():
        if not (lower[k] <= _f(v) <= upper[k]):
            inf = float("inf")
            return inf, {kk: inf for kk in theta}
    s0 = _f(theta["sigma_0"])
    A = math.exp(_f(logA))
    l0 = _f(lambda0)
    C, mask, dC = spatial_metric(theta, lower, upper, n_px_side, grad=True)
    X = _t(x)[:, mask]
    N = X.shape[0]
    XC = X @ C
    h0 = (XC * X).sum(1)
    Kvec = h0 + s0 * s0
    q = torch.sqrt(Kvec)
    qq = torch.outer(q, q)
    cosd = torch.clip((XC @ X.T + s0 * s0) / (qq + 1e-7), -1, 1)
    delta = torch.arccos(cosd)
    sind = torch.sqrt(1 - cosd * cosd)
    Kt = qq * (sind + PI32 * cosd - delta * cosd) / PI32
    Kt = (Kt + Kt.T) / 2

    Lk = torch.linalg.cholesky(Kt)
    Lv = torch.linalg.cholesky(V)
    logdetK = 2 * torch.log(torch.diagonal(Lk)).sum()
    logdetV = 2 * torch.log(torch.diagonal(Lv)).sum()
    b = torch.cholesky_solve(m[:, None], Lk)[:, 0]
    S = torch.linalg.solve_triangular(Lk, Lv, upper=False)  # L^-1 L_V  (lower triangular)
    trKinvV = (S * S).sum()
    KL = -0.5 * logdetV + 0.5 * logdetK + 0.5 * (m @ b) + 0.5 * trKinvV

    lam_m = m
    lam_var = Kvec - torch.diagonal(Kt) + torch.diagonal(V)
    f = torch.exp(A * lam_m + 0.5 * A * A * lam_var + l0)
    Lk_val = A * (r @ lam_m) + l0 * r.sum() - f.sum()
    loss = -(Lk_val - KL)
    if not want_grad:
        if want_parts:
            return float(loss), None, dict(loglik=float(Lk_val), KL=float(KL), lam_m=lam_m,
                                           lam_var=lam_var, f=f, Kt=Kt, Kvec=Kvec)
        return float(loss), None

    Kinv = torch.cholesky_inverse(Lk)
    Z = torch.linalg.solve_triangular(Lk.T, S, upper=True)  # L^-T S ; Z Z^T = K~^-1 V K~^-1
    W = 0.5 * (Kinv - Z @ Z.T - torch.outer(b, b))
    Aw = W * (PI32 - delta) / PI32
    u = (W * sind / PI32) @ q
    # diagonal correction from the likelihood term: dKvec_p,i - dK~_p,ii = h_p,i * g_i
    cd = torch.diagonal(cosd)
    dd = torch.diagonal(delta)
    Jd = (torch.diagonal(sind) + PI32 * cd - dd * cd) / PI32
    g = 1.0 - Jd - (PI32 - dd) * (1.0 - cd) / PI32
    wl = -0.5 * A * A * f * g  # dL_p = sum_i wl_i h_p,i
    # dKL_p = <dC_p, X^T Aw X> + sum_i (u_i/q_i) h_p,i   (u counted twice: rows + columns, /2 from dq)
    tvec = u / q - wl  # grad = dKL - dL
    M = X.T @ (Aw @ X) + (X * tvec[:, None]).T @ X
    grad = {k: float((dC[k] * M).sum()) for k in DC_KEYS}
    dKL_s0 = s0 * (2 * Aw.sum() + 2 * (u / q).sum())
    dL_s0 = (wl * 2 * s0).sum()
    grad["sigma_0"] = float(dKL_s0 - dL_s0)
    grad = {k: grad[k] for k in THETA_KEYS}
    if want_parts:
        return float(loss), grad, dict(loglik=float(Lk_val), KL=float(KL), lam_m=lam_m,
                                       lam_var=lam_var, f=f, Kt=Kt, Kvec=Kvec, W=W)
    return float(loss), grad


def estep_cholesky(Kt, r, m, f, logA):
    """E-step Newton update with a = I (full rank, original basis):
    s = A sqrt(f), M = I + diag(s) K~ diag(s) = L_M L_M^T, T = L_M^-1 diag(s) K~,
    V = K~ - T^T T, m_new = V (A^2 f o m + A (r - f)).  Equivalent to utils.py:1420-1438."""
    A = math.exp(_f(logA))
    s = A * torch.sqrt(f)
    M = s[:, None] * Kt * s[None, :]
    M.diagonal().add_(1.0)
    Lm = torch.linalg.cholesky(M)
    T = torch.linalg.solve_triangular(Lm, s[:, None] * Kt, upper=False)
    V = Kt - T.T @ T
    V = (V + V.T) / 2
    m_new = V @ (A * A * f * m + A * (r - f))
    return m_new, V


def predict_cholesky(theta, xstar, x, C, Kt, m, V):
    """Predictive moments in the original basis: alpha = K~^-1 k*^T,
    mu* = alpha.m, s2* = k** + alpha^T (V - K~) alpha.  Equivalent to utils.py:1486-1498."""
    ks = arccos_gram(theta, xstar, x, C)
    Lk = torch.linalg.cholesky(Kt)
    alpha = torch.cholesky_solve(ks.T.contiguous(), Lk)
    mu = alpha.T @ m
    kss = arccos_gram_diag(theta, xstar, C)
    s2 = kss + (((V - Kt) @ alpha) * alpha).sum(0)
    return mu, s2


# ------------------------------------------------------------- active-learning utility
def lambert_w0(z: torch.Tensor) -> torch.Tensor:
    """Principal branch of the Lambert W function for real z >= 0 (the reference calls
    scipy.special.lambertw(z, k=0, tol=1e-8) and keeps the real part, utils.py:464-466):
    logarithmic starting value + Fritsch's quartic iteration on w = ln(z / w)."""
    z = z.to(torch.float64)
    w = torch.where(z < 2.0, z / (1.0 + z), torch.log(z.clamp_min(2.0)) - torch.log(torch.log(z.clamp_min(2.0))))
    pos = z > 0
    zs = torch.where(pos, z, torch.ones_like(z))
    w = torch.where(pos, w, torch.ones_like(w))
    for _ in range(6):
        zn = torch.log(zs / w) - w
        q = 2.0 * (1.0 + w) * (1.0 + w + 2.0 / 3.0 * zn)
        eps = zn / (1.0 + w) * (q - zn) / (q - 2.0 * zn)
        w = w * (1.0 + eps)
    return torch.where(pos, w, torch.zeros_like(w))


def utility_terms(sigma2, mu, r):
    """p(r|x,D) of the Laplace approximation and its log for r in the given list
    (nd_lambda_r_mean + nd_p_r_given_xD, utils.py:438-496), incl. the reference's treatment of
    overflowing terms: where exp(r sigma2 + mu) sigma2 is inf, z, r sigma2, r and log r! are set
    to 0 (the term still enters the sums).  Shapes (nr, nstar)."""
    sigma2, mu, r = _t(sigma2).reshape(-1), _t(mu).reshape(-1), _t(r).reshape(-1)
    rs = torch.outer(r, sigma2)
    z = torch.exp(rs + mu) * sigma2[None, :]
    keep = z != float("inf")
    z = torch.where(keep, z, torch.zeros_like(z))
    rs = torch.where(keep, rs, torch.zeros_like(rs))
    lam = rs + mu - lambert_w0(z)                                    # utils.py:466
    e = torch.exp(lam)
    lrf = torch.where(keep, torch.lgamma(r + 1.0)[:, None].expand_as(z), torch.zeros_like(z))
    rr = torch.where(keep, r[:, None].expand_as(z), torch.zeros_like(z))
    logp = lam * rr - e - (lam - mu) ** 2 / (2.0 * sigma2[None, :]) - 0.5 * guarded_log(e * sigma2 + 1.0) - lrf  # 494
    return torch.exp(logp), logp, lrf


def active_utility(sigma2, mu, r):
    """U = H(r|x,D) - <H(r|f,x)> (nd_utility, utils.py:498-525; nd_mean_noise_entropy 413-430)."""
    sigma2, mu = _t(sigma2).reshape(-1), _t(mu).reshape(-1)
    p, logp, lrf = utility_terms(sigma2, mu, r)
    H_r = -(p * logp).sum(0)
    H_mean = -torch.exp(mu + 0.5 * sigma2) * (mu + sigma2 - 1.0) + (p * lrf).sum(0)
    return H_r - H_mean


# ------------------------------------------------------------- synthetic workload
def default_limits():
    """generate_theta's boxes (utils.py:854-855)."""
    inf = float("inf")
    lower = {"sigma_0": 0.0, "eps_0x": -1.0, "eps_0y": -1.0, "-2log2beta": -inf, "-log2rho2": -inf, "Amp": 0.0}
    upper = {"sigma_0": inf, "eps_0x": 1.0, "eps_0y": 1.0, "-2log2beta": inf, "-log2rho2": inf, "Amp": inf}
    return lower, upper
