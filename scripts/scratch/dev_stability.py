"""Accuracy of the recursive Cholesky with explicit block inverses (fit.hip:potrf_rec) against the
condition number of the matrix: factor residual, inverse residual and log-determinant error for
M = Q diag(lambda) Q^T with log-spaced eigenvalues, beside LAPACK's potrf (numpy) on the same matrices."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gaussian_processes_amd import utils as gp

def spd(n, cond, seed):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.logspace(0, -np.log10(cond), n)
    M = (Q * lam) @ Q.T
    return 0.5 * (M + M.T), lam

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eps = np.finfo(np.float64).eps
for cond in (1e2, 1e4, 1e6, 1e8, 1e10, 1e12, 1e14):
    M, lam = spd(n, cond, 3)
    Mt = torch.from_numpy(M).cuda()
    L, Li, logdet, info = gp.cholesky(Mt, want_inverse=True)
    Lh, Lih = L.cpu().numpy(), Li.cpu().numpy()
    Lh, Lih = np.tril(Lh), np.tril(Lih)
    res = np.abs(Lh @ Lh.T - M).max() / np.abs(M).max()
    inv = np.abs(Lih @ Lh - np.eye(n)).max()
    sol = np.abs(Lih.T @ Lih @ M - np.eye(n)).max()
    try:
        Lr = np.linalg.cholesky(M)
        res_ref = np.abs(Lr @ Lr.T - M).max() / np.abs(M).max()
        ld_ref = 2 * np.log(np.diag(Lr)).sum()
    except np.linalg.LinAlgError:
        res_ref, ld_ref = float("nan"), float("nan")
    ld_true = np.log(lam).sum()
    print(f"cond {cond:.0e}: info {info} |LL^T-M|/|M| {res:.2e} (LAPACK {res_ref:.2e}; eps*n {eps*n:.1e})  |Li L - I| {inv:.2e}  "
          f"|K^-1 M - I| {sol:.2e} (cond*eps {cond*eps:.1e})  logdet err {abs(logdet-ld_true):.2e} (LAPACK {abs(ld_ref-ld_true):.2e})")
