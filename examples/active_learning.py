"""The closed loop of the reference's one_cell_active_training.ipynb on synthetic stimuli: fit on a start set,
then repeatedly score every remaining image with the information-gain utility, add the best one to the training
and inducing sets and refit from the previous posterior.

    python examples/active_learning.py --pool 1200 --start 300 --iterations 5 [--notebook-step]

The step between two fits is ``utils.extend_inducing_set``: K~ grown by its latest column and, while every
eigenvalue is kept, the Cholesky factor of the previous fit extended by one row (gpfit_potrf_append) -- no
eigendecomposition per added image.  ``--notebook-step`` prepares the refit the notebook's way instead (eigh of the
grown matrix, :1889-1900) so that the two can be timed against each other; one iteration of the loop is checked
against the real reference in the GPU suite (fixture g9)."""
import argparse
import contextlib
import copy
import io
import os
import sys
import time
import warnings

import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--pool", type=int, default=1200, help="images available")
ap.add_argument("--start", type=int, default=300, help="images of the initial fit")
ap.add_argument("--iterations", type=int, default=5)
ap.add_argument("--px", type=int, default=8, help="pixels per side")
ap.add_argument("--notebook-step", action="store_true", help="eigh of the grown K~ for every added image, as the notebook does")
args = ap.parse_args()

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_processes_amd import utils as U  # noqa: E402
from gaussian_processes_amd import synthetic as syn  # noqa: E402

dev = torch.device("cuda")
d = args.px * args.px
X = torch.from_numpy(syn.stimuli(args.pool, d)).to(dev)
R = torch.from_numpy(syn.cell_inputs(args.pool, 0)[0]).to(dev)
lower, upper = syn.limits()
theta = {k: torch.tensor(float(v), dtype=torch.float64, requires_grad=True) for k, v in syn.theta0().items()}
all_idx = torch.arange(args.pool)
in_use = all_idx[: args.start]
fit_parameters = {"ntilde": args.start, "maxiter": 3, "nEstep": 2, "nMstep": 3, "nFparamstep": 3, "kernfun": "acosker",
                  "cellid": 0, "n_px_side": args.px, "display_hyper": False, "in_use_idx": in_use, "xtilde_idx": in_use}
model_args = {"fit_parameters": fit_parameters, "xtilde": X[in_use], "hyperparams_tuple": (theta, lower, upper),
              "f_params": {"logA": torch.tensor(syn.F_PARAMS["logA"]), "lambda0": torch.tensor(syn.F_PARAMS["lambda0"])}}
quiet = lambda: contextlib.redirect_stdout(io.StringIO())


def sync():
    if dev.type == "cuda":
        torch.cuda.synchronize()


with quiet(), warnings.catch_warnings():
    warnings.simplefilter("ignore")
    t0 = time.time()
    model, err = U.varGP(X[in_use], R[in_use], **model_args)
    sync()
    t_start = time.time() - t0
assert not err["is_error"], err
print(f"start fit on {args.start} images: {t_start:.2f} s, log marginal {float(model['values_track']['loss_track']['logmarginal'][-1]):.3f}")
r_counts = torch.arange(0, 100, dtype=torch.float64)
for it in range(args.iterations):
    with quiet(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t0 = time.time()
        in_use = model["fit_parameters"]["in_use_idx"]
        remaining = all_idx[~torch.isin(all_idx, in_use)]
        xtilde, xstar = X[in_use], X[remaining]
        mask, C, B = model["mask"], model["C"], model["B"]
        th, fp = model["hyperparams_tuple"][0], model["f_params"]
        A, lambda0 = torch.exp(fp["logA"]), fp["lambda0"]
        # utility of every remaining image (notebook cell 'Calculate the utility of each remaining image')
        Kvec_star = U.acosker(th, xstar[:, mask], x2=None, C=C, dC=None, diag=True)
        K_star_b = U.acosker(th, xstar[:, mask], x2=xtilde[:, mask], C=C, dC=None, diag=False) @ B
        lam_m, lam_var = U.lambda_moments(xstar[:, mask], model["K_tilde_b"], K_star_b @ model["K_tilde_inv_b"], Kvec_star,
                                          K_star_b, C, model["m_b"], model["V_b"], th)
        u = U.nd_utility(A ** 2 * lam_var, A * lam_m + lambda0, r_counts)
        best = remaining[int(u.argmax())]
        sync()
        t_score = time.time() - t0
        # add it and prepare the refit from the previous posterior
        t0 = time.time()
        in_use = torch.cat((in_use, best[None]))
        n = in_use.shape[0]
        grown = U.extend_inducing_set(model, X[best], route="eigh" if args.notebook_step else None)
        nxt = {"fit_parameters": dict(model["fit_parameters"]), "hyperparams_tuple": model["hyperparams_tuple"],
               "f_params": model["f_params"], **grown}
        nxt["fit_parameters"].update({"ntilde": n, "in_use_idx": in_use, "xtilde_idx": in_use, "maxiter": 2})
        sync()
        t_prep = time.time() - t0
        t0 = time.time()
        model, err = U.varGP(X[in_use], R[in_use], **nxt)
        sync()
        t_fit = time.time() - t0
    assert not err["is_error"], err
    print(f"iteration {it + 1}: scored {remaining.shape[0]} images in {t_score * 1e3:.1f} ms (max utility {float(u.max()):.4f}, image {int(best)}); "
          f"refit prepared in {t_prep * 1e3:.1f} ms ({nxt['init_kernel']['basis_route']} route), "
          f"refit on {n} images in {t_fit:.2f} s, log marginal {float(model['values_track']['loss_track']['logmarginal'][-1]):.3f}")
