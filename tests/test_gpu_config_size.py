"""Parity at the sizes `bench.py` times, from data the real reference produced (tests/golden/make_golden.py, run in
the build container against /root/reference; inputs regenerate from their seeds, only the reference's outputs are
stored):

* the regime the reference actually runs -- default EIGVAL_TOL (utils.py:39, 1683), where every fit from N = 1024 up
  truncates -- at N = 4096, d = 256: truncated (n_tilde = n_t; basis by subspace iteration, split-K products,
  lock-step n x n chains) and sparse (n_tilde = 2048) fused closures against the reference's closure;
* BASELINE configs[0]: `varGP` -> `test` (one_cell_fit.ipynb:384,390) at N = 512, d = 64, default tolerance;
* the grouped theta-grid route at N = 8192 (8 mixed-precision points per call) against the single path (bits) and
  the oracle; the reference-formulation oracle at N = 2048 against the fused closures;
* the fused truncated / sparse closures on freshly created contexts of minimal capacity (n_kept close to n_tilde).
"""
import contextlib
import io
import warnings

import numpy as np
import pytest
import torch

from conftest import load_golden, relerr
from gaussian_processes_amd import synthetic as syn
from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu
KEYS = syn.THETA_KEYS
LOWER, UPPER = syn.limits()
LOGA, LAM0 = syn.F_PARAMS["logA"], syn.F_PARAMS["lambda0"]


def tth(vec):
    return {k: torch.tensor(float(v), dtype=torch.float64, requires_grad=True) for k, v in zip(KEYS, vec)}


def T(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float64)).cuda()


@pytest.fixture(scope="module")
def gp():
    from gaussian_processes_amd import utils
    return utils


def fparams(logA=LOGA, lambda0=LAM0):
    return {"logA": torch.tensor(float(logA), dtype=torch.float64), "lambda0": torch.tensor(float(lambda0), dtype=torch.float64)}


def closure_inputs(gp, g, tol=None):
    """The inputs of tests/golden/make_golden.py:closure_case rebuilt from the seed on the device, the basis by this
    library's own eigen-stabilisation of K~(theta0) at the fixture's tolerance."""
    N, d, nt_ = int(g["N"]), int(g["d"]), int(g["ntilde"])
    n_px = int(g["n_px"])
    X = T(syn.stimuli(N, d, seed=int(g["seed"])))
    xt = X if nt_ == N else X[:nt_].contiguous()
    r_np, m_np = syn.cell_inputs(N)
    r, m = T(r_np), T(m_np[:nt_].copy())
    th0 = tth(g["theta0"])
    C0, mask0 = gp.localker(th0, UPPER, LOWER, n_px)
    assert bool(mask0.all())
    K0 = gp.acosker(th0, xt, xt, C=C0)
    old = gp.EIGVAL_TOL
    gp.EIGVAL_TOL = float(g["tol"]) if tol is None else tol
    try:
        _, B, _, _ = gp._stabilised_basis(K0)
        route = gp._BASIS.route
    finally:
        gp.EIGVAL_TOL = old
    m_b = gp.matmul(B, m, transA=True)
    V_b = gp.matmul(B, gp.matmul(0.5 * K0, B), transA=True)
    V_b = ((V_b + V_b.T) * 0.5).contiguous()
    return n_px, X, xt, r, B.contiguous(), m_b, V_b, route


@pytest.mark.parametrize("name,route", [("g3_closure_trunc_N4096_d256.npz", "subspace"),
                                        ("g3_closure_sparse_N4096_nt2048_d256.npz", "subspace")])
def test_fused_closures_at_config_size_match_the_reference(gp, name, route):
    """N = 4096, d = 256 at the reference's default tolerance (515 of 4096 / 534 of 2048 eigen-directions kept): the
    basis from `_stabilised_basis` (subspace iteration on the library's GEMM / Cholesky, no dense eigendecomposition, for
    the 4096-sized and the 2048-sized K~ alike), then ONE fused call -- against the real reference's closure on the same
    seeded inputs.  The value of the closure does not depend on the basis chosen inside the kept eigenspace, so the
    kept COUNT must match exactly and the loss / gradients to 1e-9 / 1e-6."""
    g = load_golden(name)
    n_px, X, xt, r, B, m_b, V_b, took = closure_inputs(gp, g)
    assert took == route
    assert B.shape[1] == int(g["n_kept"]), (B.shape, int(g["n_kept"]))
    th = tth(g["theta"])
    fp = fparams(g["logA"], g["lambda0"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if xt is X:
            loss, grad = gp._closure_projected(th, (LOWER, UPPER), n_px, X, r, B, m_b, V_b, fp)
        else:
            loss, grad = gp._closure_sparse(th, (LOWER, UPPER), n_px, X, xt, r, B, m_b, V_b, fp)
    d_loss = abs(loss - float(g["loss"])) / abs(float(g["loss"]))
    gv = np.array([grad[k] for k in KEYS])
    d_grad = np.abs(gv - g["grad"]).max() / np.abs(g["grad"]).max()
    print(f"{name}: kept {B.shape[1]}, loss {d_loss:.2e}, grad {d_grad:.2e}")
    assert d_loss <= 1e-9, d_loss
    assert d_grad <= 1e-6, d_grad


def test_reference_formulation_oracle_at_N2048_against_the_fused_closures(gp):
    """`oracle.mstep_closure_reference(tol=1e-4)` -- the reference's own op sequence (materialised dK, projections, LU
    inverse; pinned to the reference by tests/test_oracle_golden.py) -- at N = 2048, d = 256 against
    `gpfit_fit_eval_projected` (n_tilde = n_t) and `gpfit_fit_eval_sparse` (n_tilde = 1024) on the same B, m_b, V_b."""
    N, d, n_px = 2048, 256, 16
    X = T(syn.stimuli(N, d))
    r_np, m_np = syn.cell_inputs(N)
    r = T(r_np)
    th0, th1 = tth([syn.theta0()[k] for k in KEYS]), tth([syn.theta_eval()[k] for k in KEYS])
    C0, _ = gp.localker(th0, UPPER, LOWER, n_px)
    fp = fparams()
    for nt_ in (N, 1024):
        xt = X if nt_ == N else X[:nt_].contiguous()
        K0 = gp.acosker(th0, xt, xt, C=C0)
        _, B, _, _ = gp._stabilised_basis(K0)
        assert 64 < B.shape[1] < nt_
        m_b = gp.matmul(B, T(m_np[:nt_].copy()), transA=True)
        V_b = gp.matmul(B, gp.matmul(0.5 * K0, B), transA=True)
        V_b = ((V_b + V_b.T) * 0.5).contiguous()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if nt_ == N:
                loss, grad = gp._closure_projected(th1, (LOWER, UPPER), n_px, X, r, B, m_b, V_b, fp)
            else:
                loss, grad = gp._closure_sparse(th1, (LOWER, UPPER), n_px, X, xt, r, B, m_b, V_b, fp)
            ref_loss, ref_grad = orc.mstep_closure_reference(syn.theta_eval(), LOWER, UPPER, n_px, X.cpu(), xt.cpu(), r.cpu(),
                                                             B.cpu(), m_b.cpu(), V_b.cpu(), LOGA, LAM0, tol=1e-4)
        a, b = np.array([grad[k] for k in KEYS]), np.array([ref_grad[k] for k in KEYS])
        assert abs(loss - ref_loss) <= 1e-9 * abs(ref_loss), (nt_, loss, ref_loss)
        assert np.abs(a - b).max() <= 1e-6 * np.abs(b).max(), (nt_, a, b)


def test_vargp_config0_N512_matches_reference(gp):
    """BASELINE configs[0] = the call pattern of one_cell_fit.ipynb:384,390 at N = 512, d = 64 with the reference's
    default EIGVAL_TOL -- the size where its truncation rule keeps everything by a 2 % margin, so
    the rank decision is re-taken close to its threshold after every M-step.  The real reference kept 512 of 512 at
    every tracked iteration; tracks, final theta / logA, the posterior in the original basis and both predictions
    (`at_iteration=None` and `=2`) must follow."""
    g = load_golden("g6_vargp_config0_N512.npz")
    N, d = int(g["N"]), int(g["d"])
    X = T(syn.stimuli(N, d, seed=0))
    r = T(syn.cell_inputs(N)[0])
    nt_ = int(g["ntilde"])
    fit_parameters = {"ntilde": nt_, "maxiter": int(g["maxiter"]), "nEstep": int(g["nEstep"]), "nMstep": int(g["nMstep"]),
                      "nFparamstep": int(g["nFparamstep"]), "kernfun": "acosker", "cellid": 0, "n_px_side": 8,
                      "display_hyper": False}
    args = {"fit_parameters": fit_parameters, "xtilde": X[:nt_].clone(), "hyperparams_tuple": (tth(g["theta0"]), LOWER, UPPER),
            "f_params": fparams()}
    assert gp.EIGVAL_TOL == float(g["tol"]) == 1e-4
    Rt = T(np.random.default_rng(5).poisson(0.7, (4, 6, 1)).astype(np.float64))
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fit, err = gp.varGP(X, r, **args)
        _, R_pred, _, _ = gp.test(T(g["Xstar"]), Rt, X_train=X, at_iteration=None, **fit)
        _, R_pred2, _, _ = gp.test(T(g["Xstar"]), Rt, X_train=X, at_iteration=2, **fit)
    assert not err["is_error"], err
    vt = fit["values_track"]
    kept = [int(v.shape[0]) for v in vt["variation_par_track"]["V_b"]]
    assert kept == [int(k) for k in g["n_kept_track"]], kept
    assert fit["B"].shape[1] == int(g["n_kept"])
    d_track = relerr(vt["loss_track"]["logmarginal"].numpy(), g["logmarginal"])
    d_ll = relerr(vt["loss_track"]["loglikelihood"].numpy(), g["loglikelihood"])
    d_kl = relerr(vt["loss_track"]["KL"].numpy(), g["KL"])
    th_final = np.array([float(fit["hyperparams_tuple"][0][k]) for k in KEYS])
    d_theta = float(np.abs(th_final - g["theta_final"]).max())
    d_logA = abs(float(fit["f_params"]["logA"]) - float(g["logA_final"]))
    B = fit["B"]
    m_orig = gp.matmul(B, fit["m_b"])
    V_orig = gp.matmul(gp.matmul(B, fit["V_b"]), B, transB=True)
    probe = T(np.random.default_rng(int(g["probe_seed"])).standard_normal(nt_))
    d_m = relerr(m_orig.cpu().numpy(), g["m_orig"])
    d_vd = relerr(torch.diagonal(V_orig).cpu().numpy(), g["V_orig_diag"])
    d_vp = relerr(gp.matmul(V_orig, probe).cpu().numpy(), g["V_orig_probe"])
    d_p, d_p2 = relerr(R_pred.cpu().numpy(), g["R_pred"]), relerr(R_pred2.cpu().numpy(), g["R_pred_it2"])
    print(f"config0: tracks {d_track:.2e} / {d_ll:.2e} / {d_kl:.2e}, theta {d_theta:.2e}, logA {d_logA:.2e}, m {d_m:.2e}, "
          f"diag V {d_vd:.2e}, V probe {d_vp:.2e}, predictions {d_p:.2e} / {d_p2:.2e}")
    assert d_track < 1e-6 and d_ll < 1e-6 and d_kl < 1e-5, (d_track, d_ll, d_kl)
    assert d_theta < 1e-4 and d_logA < 1e-4, (d_theta, d_logA)
    assert d_m < 1e-4 and d_vd < 1e-4 and d_vp < 1e-4, (d_m, d_vd, d_vp)
    assert d_p < 1e-4 and d_p2 < 1e-4, (d_p, d_p2)


def test_theta_grid_group_of_8_at_N8192_equals_single_path_and_oracle():
    """BASELINE configs[4] as `bench.py --config thetagrid --group 8` runs it: 8 mixed-precision theta points of one cell
    at N = 8192, d = 256 per `gpfit_fit_eval_batch` call, the second group reusing every context's V factor.  Every
    output scalar of the 16 points equals, bit for bit, the same point evaluated alone on one context; two points are
    compared with the fp64 oracle (the mixed mode's measured distance from fp64 is 2e-9 on the loss, 3e-6 on the
    gradients; north star 1e-5)."""
    from gaussian_processes_amd.engine import GPFitEngine, fit_eval_group
    from test_gpu_group import key
    import bench
    N, d, group = 8192, 256, 8
    dev = torch.device("cuda:0")
    grid = syn.grid_for(d)
    Xh = torch.from_numpy(syn.stimuli(N, d))
    X = Xh.to(dev)
    r_np, m_np = syn.cell_inputs(N, 0)
    r, m = torch.from_numpy(r_np).to(dev), torch.from_numpy(m_np).to(dev)
    V = bench.build_V(X, grid, syn.theta0(), dev)
    pts = syn.theta_grid(8)
    thetas = [pts[(37 * u + 11) % len(pts)] for u in range(2 * group)]
    engs = [GPFitEngine(N, d) for _ in range(group)]
    try:
        first = fit_eval_group(engs, thetas[:group], LOWER, UPPER, grid, X, r, m, V, LOGA, LAM0, grad_precision="f32")
        second = fit_eval_group(engs, thetas[group:], LOWER, UPPER, grid, X, r, m, V, LOGA, LAM0, grad_precision="f32",
                                reuse_V=True)
        alone = [engs[0].fit_eval(th, LOWER, UPPER, grid, X, r, m, V, LOGA, LAM0, want_vectors=False, grad_precision="f32",
                                  reuse_V=True) for th in thetas]
    finally:
        for e in engs:
            e.close()
    assert [key(a) for a in alone] == [key(g) for g in first + second]
    assert len({a["loss"] for a in alone}) == 2 * group
    Vh = V.cpu()
    for u in (3, 12):
        loss, grad = orc.mstep_closure_cholesky(thetas[u], LOWER, UPPER, grid, Xh, r.cpu(), m.cpu(), Vh, LOGA, LAM0)
        got = (first + second)[u]
        ref = np.array([grad[k] for k in KEYS]); gv = np.array([got["grad"][k] for k in KEYS])
        assert abs(got["loss"] - loss) <= 1e-7 * abs(loss), (u, got["loss"], loss)
        assert np.abs(gv - ref).max() <= 1e-5 * np.abs(ref).max(), (u, gv, ref)


# ---------------------------------------------------------------- contexts of minimal capacity
@contextlib.contextmanager
def fresh_pool(gp):
    """The drop-in module's per-thread context pool replaced by an empty one: the next fused call allocates a context
    sized exactly for ITS problem (`_closure_sparse`: max(n_t, n_tilde)), not one an earlier test has grown."""
    old = gp._POOL
    gp._POOL = gp._EnginePool()
    try:
        yield
    finally:
        gp._POOL = old


def test_sparse_fixture_on_a_context_of_exactly_its_size(gp):
    """n_t = 96, n_tilde = 40, all 40 directions kept: np_cap = nb = 128 -- every n x n work matrix of the V_b chain
    needs the context's full capacity (the four-slots-in-one-buffer layout of round 3 overflowed here)."""
    g = load_golden("g3_closure_sparse_N96_nt40.npz")
    X, r, B, m_b, V_b = T(g["X"]), T(g["r"]), T(g["B"]), T(g["m_b"]), T(g["V_b"])
    xt = X[: int(g["ntilde"])].contiguous()
    fp = fparams(g["logA"], g["lambda0"])
    with fresh_pool(gp), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        loss, grad = gp._closure_sparse(tth(g["theta"]), (LOWER, UPPER), int(g["n_px"]), X, xt, r, B, m_b, V_b, fp)
        assert gp.get_engine(1, 1).n_max == 96          # the context really was sized for this problem
    assert abs(loss - float(g["loss"])) <= 1e-10 * abs(float(g["loss"]))
    assert np.abs(np.array([grad[k] for k in KEYS]) - g["grad"]).max() <= 1e-8 * np.abs(g["grad"]).max()


@pytest.mark.parametrize("nt,ntilde,tol", [(1536, 1024, 1e-7), (1024, 1024, 1e-7), (640, 512, 1e-14), (1300, 1300, 1e-14)])
def test_fused_closures_on_tight_contexts_with_most_directions_kept(gp, nt, ntilde, tol):
    """Fused sparse / truncated closures on a freshly created context of capacity max(n_t, n_tilde) with n_kept close
    to (or equal to) n_tilde -- round_up(n_kept, 128) > np_cap / 2 and = np_cap -- against the step-by-step
    formulations they fuse (which run on library primitives and torch)."""
    d, n_px = 64, 8
    X = T(syn.stimuli(nt, d, seed=5))
    xt = X if ntilde == nt else X[:ntilde].contiguous()
    r_np, m_np = syn.cell_inputs(nt)
    r = T(r_np)
    th0, th1 = tth([syn.theta0()[k] for k in KEYS]), tth([syn.theta_eval()[k] for k in KEYS])
    C0, _ = gp.localker(th0, UPPER, LOWER, n_px)
    K0 = gp.acosker(th0, xt, xt, C=C0)
    ev, evec = torch.linalg.eigh(K0)
    keep = ev > max(float(ev.max()) * tol, tol)
    B = evec[:, keep].contiguous()
    nb, cap = -(-B.shape[1] // 128) * 128, -(-max(nt, ntilde) // 128) * 128
    assert 2 * nb > cap, (B.shape, cap)
    m_b = gp.matmul(B, T(m_np[:ntilde].copy()), transA=True)
    V_b = (0.5 * torch.diag(ev[keep])).contiguous()
    fp = fparams()
    lims = (LOWER, UPPER)
    with fresh_pool(gp), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if ntilde == nt:
            l_f, g_f = gp._closure_projected(th1, lims, n_px, X, r, B, m_b, V_b, fp)
        else:
            l_f, g_f = gp._closure_sparse(th1, lims, n_px, X, xt, r, B, m_b, V_b, fp)
        assert gp.get_engine(1, 1).n_max == max(nt, ntilde)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if ntilde == nt:
            l_s, g_s = gp._closure_projected_steps(th1, lims, n_px, X, r, B, m_b, V_b, fp)
        else:
            l_s, g_s = gp._closure_sparse_steps(th1, lims, n_px, X, xt, r, B, m_b, V_b, fp)
    a, b = np.array([g_f[k] for k in KEYS]), np.array([g_s[k] for k in KEYS])
    assert abs(l_f - l_s) <= 1e-10 * abs(l_s), (l_f, l_s)
    assert np.abs(a - b).max() <= 1e-8 * np.abs(b).max(), (a, b)


@pytest.mark.parametrize("nt,nb", [(300, 300), (640, 385), (1000, 128)])
def test_projected_estep_on_a_tight_context(gp, nt, nb):
    """gpfit_estep_projected on a freshly created context of capacity max(n_t, n_kept): every padded work matrix
    (rows round_up(n_t, 128), columns round_up(n_kept, 128)) at or near the context's capacity, against the
    product-by-product update and, with the moments, against lambda_moments."""
    rng = np.random.default_rng(11)
    Mx = rng.standard_normal((nb, nb + 40))
    Ktb = T(Mx @ Mx.T / (nb + 40) + 0.05 * np.eye(nb))
    a = T(rng.standard_normal((nt, nb)) / np.sqrt(nb))
    m_b = T(0.2 * rng.standard_normal(nb))
    f = T(np.exp(0.3 * rng.standard_normal(nt)))
    r = T(rng.poisson(1.0, nt).astype(np.float64))
    Kvec = T(2.0 + rng.random(nt))
    Kb = gp.matmul(a, Ktb)
    fp = {"logA": torch.tensor(float(np.log(0.5)), dtype=torch.float64)}
    Lb, _, _, info = gp.cholesky(Ktb)
    assert info == 0
    aL = gp.matmul(a, Lb)
    m_s, V_s = gp._estep_given_factor(r, a, m_b, fp, f, Lb)
    lm_s, lv_s = gp.lambda_moments(None, Ktb, a, Kvec, Kb, None, m_s, V_s, None)
    with fresh_pool(gp):
        m_f, V_f, lm, lv = gp._estep_projected(r, a, aL, Lb, m_b, fp, f, kv0=Kvec - torch.sum(Kb * a, 1))
        assert gp.get_engine(1, 1).n_max == max(nt, nb)
    assert relerr(m_f.cpu().numpy(), m_s.cpu().numpy()) < 1e-10
    assert relerr(V_f.cpu().numpy(), V_s.cpu().numpy()) < 1e-10
    assert relerr(lm.cpu().numpy(), lm_s.cpu().numpy()) < 1e-10
    assert relerr(lv.cpu().numpy(), lv_s.cpu().numpy()) < 1e-10


@pytest.mark.parametrize("fixture", ["g6_vargp_trunc_N4096.npz", "g6_vargp_trunc_N1536.npz", "g6_vargp_trunc_N1024.npz",
                                     "g6_vargp_sparse_N3160_nt2100.npz", "g6_vargp_sparse_N2000_nt1200.npz"])
def test_vargp_default_tolerance_whole_fits_match_reference(gp, fixture):
    """(N = 1536 and 1024: the same with the kept eigenspace from the spectral projector of K~ itself,
    `eigtop.kept_eigenspace_dense`, the route below N = 1792 -- 552 of 1536 and 567 of 1024 directions kept.
    sparse_N3160_nt2100: the lab's shape (one_cell_fit.ipynb:89) in the sparse regime -- warm-started sweeps, the fused
    sparse closure, the fused projected E-step; sparse_N2000_nt1200: the same with the projector route.)
    A whole EM fit of the REAL reference at N = 4096 (d = 64, default EIGVAL_TOL: every iteration truncates) against
    the drop-in `varGP` -> `test`, whose basis at this size comes from the subspace solver without any dense
    eigendecomposition (`basis_route == 'subspace'`: the columns of B are not the reference's eigenvectors, so the posterior
    is compared in the ORIGINAL basis): kept count per tracked iteration equal, log-marginal / log-likelihood / KL tracks,
    final theta and logA, the posterior mean, the diagonal of its covariance and the covariance applied to a probe vector,
    and both predictions (`at_iteration=None` and `=2`, the latter through a rebuilt basis)."""
    g = load_golden(fixture)
    N, d = int(g["N"]), int(g["d"])
    X = T(syn.stimuli(N, d, seed=0))
    r = T(syn.cell_inputs(N)[0])
    nt_ = int(g["ntilde"])
    fit_parameters = {"ntilde": nt_, "maxiter": int(g["maxiter"]), "nEstep": int(g["nEstep"]), "nMstep": int(g["nMstep"]),
                      "nFparamstep": int(g["nFparamstep"]), "kernfun": "acosker", "cellid": 0, "n_px_side": 8,
                      "display_hyper": False}
    args = {"fit_parameters": fit_parameters, "xtilde": X[:nt_].clone(), "hyperparams_tuple": (tth(g["theta0"]), LOWER, UPPER),
            "f_params": fparams()}
    assert gp.EIGVAL_TOL == float(g["tol"]) == 1e-4
    Rt = T(np.random.default_rng(5).poisson(0.7, (4, 6, 1)).astype(np.float64))
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fit, err = gp.varGP(X, r, **args)
        _, R_pred, _, _ = gp.test(T(g["Xstar"]), Rt, X_train=X, at_iteration=None, **fit)
        _, R_pred2, _, _ = gp.test(T(g["Xstar"]), Rt, X_train=X, at_iteration=2, **fit)
    assert not err["is_error"], err
    vt = fit["values_track"]
    kept = [int(v.shape[0]) for v in vt["variation_par_track"]["V_b"]]
    assert kept == [int(k) for k in g["n_kept_track"]], (kept, g["n_kept_track"])
    assert set(vt["variation_par_track"]["basis_route"]) == {"subspace"} and fit["basis_route"] == "subspace"
    d_track = relerr(vt["loss_track"]["logmarginal"].numpy(), g["logmarginal"])
    d_ll = relerr(vt["loss_track"]["loglikelihood"].numpy(), g["loglikelihood"])
    d_kl = relerr(vt["loss_track"]["KL"].numpy(), g["KL"])
    th_final = np.array([float(fit["hyperparams_tuple"][0][k]) for k in KEYS])
    d_theta = float(np.abs(th_final - g["theta_final"]).max())
    d_logA = abs(float(fit["f_params"]["logA"]) - float(g["logA_final"]))
    B = fit["B"]
    m_orig = gp.matmul(B, fit["m_b"])
    V_orig = gp.matmul(gp.matmul(B, fit["V_b"]), B, transB=True)
    probe = T(np.random.default_rng(int(g["probe_seed"])).standard_normal(nt_))
    d_m = relerr(m_orig.cpu().numpy(), g["m_orig"])
    d_vd = relerr(torch.diagonal(V_orig).cpu().numpy(), g["V_orig_diag"])
    d_vp = relerr(gp.matmul(V_orig, probe).cpu().numpy(), g["V_orig_probe"])
    d_p, d_p2 = relerr(R_pred.cpu().numpy(), g["R_pred"]), relerr(R_pred2.cpu().numpy(), g["R_pred_it2"])
    print(f"N={N} n_tilde={nt_} whole fit: kept {kept}, tracks {d_track:.2e} / {d_ll:.2e} / {d_kl:.2e}, theta {d_theta:.2e}, logA {d_logA:.2e}, "
          f"m {d_m:.2e}, diag V {d_vd:.2e}, V probe {d_vp:.2e}, predictions {d_p:.2e} / {d_p2:.2e}")
    # (measured: tracks 2e-14 ... 1e-11, theta 5e-13, logA 3e-10, posterior 4e-10, predictions 2e-10)
    assert d_track < 1e-8 and d_ll < 1e-8 and d_kl < 1e-7, (d_track, d_ll, d_kl)
    assert d_theta < 1e-7 and d_logA < 1e-6, (d_theta, d_logA)
    assert d_m < 1e-6 and d_vd < 1e-6 and d_vp < 1e-6, (d_m, d_vd, d_vp)
    assert d_p < 1e-6 and d_p2 < 1e-6, (d_p, d_p2)
