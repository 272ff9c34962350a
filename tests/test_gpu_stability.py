"""Numerical stability of the recursive Cholesky whose triangular solves are GEMMs against explicit
block inverses (fit.hip:potrf_rec) -- the VERDICT r01 robustness item: 'only conditionally stable ...
unexamined near the reference's keep-all margin'.  The reference keeps all eigen-directions while
cond(K~) < 1e4 (EIGVAL_TOL = 1e-4, utils.py:1683) and its fixtures' full-rank family runs at tol 1e-14,
so the range examined is cond = 1e2 .. 1e14.

Measured on MI355X (scripts/scratch/dev_stability.py, N = 1024): |L L^T - M| / |M| = 1e-15 .. 2e-15 at every
condition number (LAPACK's potrf on the same matrices: 0.4e-15 .. 1.2e-15), |L^-1 L - I| follows
eps * sqrt(cond) (1e-15 at 1e2, 1.5e-9 at 1e14), log|M| errors equal LAPACK's (both limited by
cond * eps).  Asserted with a factor of ~10 of slack."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps


def spd_with_condition(n, cond, seed):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.logspace(0, -np.log10(cond), n)
    M = (Q * lam) @ Q.T
    return 0.5 * (M + M.T), lam


@pytest.mark.parametrize("n", [640, 1024])          # 5 leaves (ragged recursion) and 8 leaves
@pytest.mark.parametrize("cond", [1e2, 1e4, 1e8, 1e12, 1e14])
def test_factor_is_backward_stable_at_any_condition_number(n, cond):
    from gaussian_processes_amd import utils as gp
    M, lam = spd_with_condition(n, cond, 3)
    L, Li, logdet, info = gp.cholesky(torch.from_numpy(M).cuda(), want_inverse=True)
    assert info == 0
    L, Li = np.tril(L.cpu().numpy()), np.tril(Li.cpu().numpy())
    assert np.abs(L @ L.T - M).max() <= 2e-14 * np.abs(M).max()                 # independent of cond
    assert np.abs(Li @ L - np.eye(n)).max() <= 20 * EPS * np.sqrt(cond) + 1e-13  # ~ eps * cond(L)
    # log-determinant: as accurate as LAPACK's factor allows
    Lr = np.linalg.cholesky(M)
    ld_true = np.log(lam).sum()
    assert abs(logdet - ld_true) <= 10 * abs(2 * np.log(np.diag(Lr)).sum() - ld_true) + 1e-10 * abs(ld_true) + 1e-9


def test_unit_of_work_on_an_ill_conditioned_kernel_matrix():
    """The fused evaluation on a kernel matrix far more ill-conditioned than the bench inputs: half of
    the stimuli are near-duplicates (1e-3 apart) of the other half, so K~ has 256 eigenvalues ~1e6 times
    smaller than the rest.  Loss and gradients still match the oracle's LAPACK-based Cholesky formulation;
    tolerances widened by what cond * eps allows (loss 1e-8, gradients 1e-5 of the largest)."""
    from gaussian_processes_amd import synthetic as syn
    from gaussian_processes_amd.engine import GPFitEngine
    from oracle import gp_oracle as orc
    N, d = 512, 64
    dev = torch.device("cuda:0")
    grid = syn.grid_for(d)
    lower, upper = syn.limits()
    Xn = syn.stimuli(N, d)
    rng = np.random.default_rng(11)
    Xn[256:] = Xn[:256] + 1e-3 * rng.standard_normal((256, d))
    X = torch.from_numpy(Xn)
    r_np, m_np = syn.cell_inputs(N)
    r, m = torch.from_numpy(r_np), torch.from_numpy(m_np)
    th0, th1 = syn.theta0(), syn.theta_eval()
    C0, mask0 = orc.spatial_metric(th0, lower, upper, grid)
    K0 = orc.arccos_gram(th0, X, X, C0)
    w = np.linalg.eigvalsh(K0.numpy())
    cond = w[-1] / w[0]
    assert 1e7 < cond < 1e12, cond
    V = 0.5 * K0
    logA, lam0 = syn.F_PARAMS["logA"], syn.F_PARAMS["lambda0"]
    loss, grad, p = orc.mstep_closure_cholesky(th1, lower, upper, grid, X, r, m, V, logA, lam0, want_parts=True)
    eng = GPFitEngine(N, d)
    out = eng.fit_eval(th1, lower, upper, grid, X.to(dev), r.to(dev), m.to(dev), V.to(dev), logA, lam0)
    eng.close()
    print(f"cond(K~) = {cond:.2e}: loss rel {abs(out['loss'] - loss) / abs(loss):.2e}")
    assert abs(out["loss"] - loss) <= 1e-8 * abs(loss), (out["loss"], loss, cond)
    assert abs(out["KL"] - p["KL"]) <= 1e-8 * abs(p["KL"])
    g = np.array([out["grad"][k] for k in syn.THETA_KEYS]); gr = np.array([grad[k] for k in syn.THETA_KEYS])
    assert np.abs(g - gr).max() <= 1e-5 * np.abs(gr).max(), (g, gr, cond)
