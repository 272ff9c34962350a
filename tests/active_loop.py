"""One iteration of the closed loop of the reference's one_cell_active_training.ipynb, written
against any module with the reference's ``utils`` surface.  Used twice: by
tests/golden/make_golden.py with the real reference (fixture g9_active_step.npz) and by
tests/test_gpu_dropin.py with gaussian_processes_amd.utils -- the same statements, so the test
reads like the notebook."""
import copy

import torch

from gaussian_processes_amd import synthetic as syn

KEYS = syn.THETA_KEYS


def tth(th):
    return {k: torch.tensor(float(v), dtype=torch.float64, requires_grad=True) for k, v in th.items()}


def active_loop_step(U, X, R, n_start, maxiter, dev=None):
    """One iteration of the closed loop of one_cell_active_training.ipynb (cells 'Calculate the
    utility of each remaining image' ... 'Fit new model') written against a utils-like module
    ``U``: initial fit on the first n_start images, utility of every remaining image, the best one
    appended as training + inducing point with the kernel matrices updated by their latest column,
    refit from (m, V, init_kernel).  Shared by the golden generator (U = the reference) and the
    GPU test (U = gaussian_processes_amd.utils)."""
    lower, upper = syn.limits()
    tt = (lambda a: a.to(dev)) if dev is not None else (lambda a: a)
    X, R = tt(X), tt(R)
    all_idx = torch.arange(X.shape[0])
    in_use_idx = all_idx[:n_start]
    theta = tth(syn.theta0())
    fit_parameters = {"ntilde": n_start, "maxiter": maxiter, "nEstep": 2, "nMstep": 3, "nFparamstep": 3,
                      "kernfun": "acosker", "cellid": 0, "n_px_side": 8, "display_hyper": False,
                      "in_use_idx": in_use_idx, "xtilde_idx": in_use_idx}
    init_model = {"fit_parameters": fit_parameters, "xtilde": X[in_use_idx], "hyperparams_tuple": (theta, lower, upper),
                  "f_params": {"logA": torch.tensor(syn.F_PARAMS["logA"]), "lambda0": torch.tensor(syn.F_PARAMS["lambda0"])}}
    start_model, err = U.varGP(X[in_use_idx], R[in_use_idx], **init_model)
    assert not err["is_error"], err
    active_model = copy.deepcopy(start_model)
    # ---- retrieve (notebook region 'Retreive the values from the last model fit')
    in_use_idx = active_model["fit_parameters"]["in_use_idx"]
    xtilde_idx = active_model["fit_parameters"]["xtilde_idx"]
    remaining_idx = all_idx[~torch.isin(all_idx, in_use_idx)]
    xtilde = X[xtilde_idx]
    xstar = X[remaining_idx]
    kernfun = U.acosker
    final_kernel = active_model["final_kernel"]
    mask, C, B = active_model["mask"], active_model["C"], active_model["B"]
    K_tilde_b, K_tilde_inv_b = active_model["K_tilde_b"], active_model["K_tilde_inv_b"]
    m_b, V_b, f_params = active_model["m_b"], active_model["V_b"], active_model["f_params"]
    theta = active_model["hyperparams_tuple"][0]
    A = torch.exp(f_params["logA"])
    lambda0 = f_params["lambda0"]
    # ---- utility of each remaining image
    Kvec_star = kernfun(theta, xstar[:, mask], x2=None, C=C, dC=None, diag=True)
    K_star = kernfun(theta, xstar[:, mask], x2=xtilde[:, mask], C=C, dC=None, diag=False)
    K_star_b = K_star @ B
    lam_m, lam_var = U.lambda_moments(xstar[:, mask], K_tilde_b, K_star_b @ K_tilde_inv_b, Kvec_star, K_star_b, C, m_b, V_b, theta)
    logf_mean = A * lam_m + lambda0
    logf_var = A ** 2 * lam_var
    r_masked = torch.arange(0, 100, dtype=torch.float64)
    u2d = U.nd_utility(logf_var, logf_mean, r_masked)
    i_best = u2d.argmax()
    x_idx_best = remaining_idx[int(i_best)]
    # ---- update indices and kernels
    in_use_idx = torch.cat((in_use_idx, x_idx_best[None]))
    xtilde_idx = in_use_idx
    ntilde = xtilde_idx.shape[0]
    X_in_use, R_in_use = X[in_use_idx], R[in_use_idx]
    xtilde_updated = X[xtilde_idx]
    active_model["xtilde"] = xtilde_updated
    active_model["fit_parameters"]["ntilde"] = ntilde
    active_model["fit_parameters"]["in_use_idx"] = in_use_idx
    active_model["fit_parameters"]["xtilde_idx"] = xtilde_idx
    V = B @ V_b @ B.T
    V = 0.5 * (V + V.T)
    m = B @ m_b
    V_new = torch.eye(ntilde, dtype=V_b.dtype, device=V_b.device)
    V_new[: ntilde - 1, : ntilde - 1] = V
    active_model["V"] = V_new
    active_model["m"] = torch.cat((m, m.mean()[None]))
    K_tilde_reduced = final_kernel["K_tilde"]
    K_tilde_column = kernfun(theta, xtilde_updated[:, mask], xtilde_updated[-1, mask][None], C=C, dC=None, diag=False)
    K_tilde = torch.cat((K_tilde_reduced, K_tilde_column[:-1]), axis=1)
    K_tilde = torch.cat((K_tilde, K_tilde_column.T), axis=0)
    K = K_tilde
    Kvec = kernfun(theta, X_in_use[:, mask], x2=None, C=C, dC=None, diag=True)
    eigvals, eigvecs = torch.linalg.eigh(K_tilde, UPLO="L")
    ikeep = eigvals > max(eigvals.max() * U.EIGVAL_TOL, U.EIGVAL_TOL)
    Bn = eigvecs[:, ikeep]
    init_kernel = {"C": C, "mask": mask, "K_tilde": K_tilde, "K": K, "Kvec": Kvec, "B": Bn,
                   "K_tilde_b": torch.diag(eigvals[ikeep]), "K_b": K @ Bn,
                   "K_tilde_inv_b": torch.diag_embed(1 / eigvals[ikeep]), "KKtilde_inv_b": Bn}
    active_model["init_kernel"] = init_kernel
    refit, err = U.varGP(X_in_use, R_in_use, **active_model)
    assert not err["is_error"], err
    return {"u2d": u2d, "i_best": int(i_best), "x_idx_best": int(x_idx_best), "K_tilde_new": K_tilde,
            "start_logmarginal": start_model["values_track"]["loss_track"]["logmarginal"],
            "refit_logmarginal": refit["values_track"]["loss_track"]["logmarginal"],
            "refit_theta": torch.tensor([float(refit["hyperparams_tuple"][0][k]) for k in KEYS]),
            "refit_logA": float(refit["f_params"]["logA"]), "n_kept": int(Bn.shape[1])}
