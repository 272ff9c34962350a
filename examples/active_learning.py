"""The closed loop of the reference's one_cell_active_training.ipynb on synthetic stimuli: fit on a start set,
then repeatedly score every remaining image with the information-gain utility, add the best one as training +
inducing point, extend the kernel matrices by their latest column and refit from the previous (m, V).

    python examples/active_learning.py --pool 1200 --start 300 --iterations 5

The statements are those of tests/active_loop.py (one iteration, checked against the real reference in the GPU
suite, fixture g9); here they run in a loop with timings.  Works with any module that has the reference's utils
surface (--utils /path/to/Spatial_GP_repo runs the reference itself on the CPU)."""
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
ap.add_argument("--utils", default=None)
args = ap.parse_args()

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if args.utils:
    sys.path.insert(0, args.utils)
    sys.dont_write_bytecode = True
    os.environ.setdefault("MPLBACKEND", "Agg")
    with contextlib.redirect_stdout(io.StringIO()):
        import utils as U
else:
    from gaussian_processes_amd import utils as U
from gaussian_processes_amd import synthetic as syn  # noqa: E402

dev = torch.device("cuda") if (not args.utils and torch.cuda.is_available()) else torch.device("cpu")
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
        # add it, extend the kernel matrices by their latest column, refit from the previous (m, V)
        t0 = time.time()
        nxt = copy.deepcopy(model)
        in_use = torch.cat((in_use, best[None]))
        n = in_use.shape[0]
        nxt["xtilde"] = X[in_use]
        nxt["fit_parameters"].update({"ntilde": n, "in_use_idx": in_use, "xtilde_idx": in_use, "maxiter": 2})
        V = B @ model["V_b"] @ B.T
        V_new = torch.eye(n, dtype=V.dtype, device=V.device)
        V_new[: n - 1, : n - 1] = 0.5 * (V + V.T)
        m = B @ model["m_b"]
        nxt["V"], nxt["m"] = V_new, torch.cat((m, m.mean()[None]))
        col = U.acosker(th, X[in_use][:, mask], X[in_use][-1, mask][None], C=C, dC=None, diag=False)
        K_tilde = torch.cat((torch.cat((model["final_kernel"]["K_tilde"], col[:-1]), axis=1), col.T), axis=0)
        Kvec = U.acosker(th, X[in_use][:, mask], x2=None, C=C, dC=None, diag=True)
        ev, evec = torch.linalg.eigh(K_tilde, UPLO="L")
        keep = ev > max(ev.max() * U.EIGVAL_TOL, U.EIGVAL_TOL)
        Bn = evec[:, keep]
        nxt["init_kernel"] = {"C": C, "mask": mask, "K_tilde": K_tilde, "K": K_tilde, "Kvec": Kvec, "B": Bn,
                              "K_tilde_b": torch.diag(ev[keep]), "K_b": K_tilde @ Bn,
                              "K_tilde_inv_b": torch.diag_embed(1 / ev[keep]), "KKtilde_inv_b": Bn}
        model, err = U.varGP(X[in_use], R[in_use], **nxt)
        sync()
        t_fit = time.time() - t0
    assert not err["is_error"], err
    print(f"iteration {it + 1}: scored {remaining.shape[0]} images in {t_score * 1e3:.1f} ms (max utility {float(u.max()):.4f}, image {int(best)}); "
          f"refit on {n} images in {t_fit:.2f} s, log marginal {float(model['values_track']['loss_track']['logmarginal'][-1]):.3f}")
