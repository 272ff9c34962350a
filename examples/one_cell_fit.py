"""The flow of the reference's one_cell_fit.ipynb on synthetic stimuli: hyperparameter set-up,
varGP (EM fit: E-steps, firing-rate parameters, L-BFGS M-steps), test() on held-out images.

    python examples/one_cell_fit.py --n 512 --d 64            # this library, on the MI355X
    python examples/one_cell_fit.py --n 512 --d 64 --utils /path/to/Spatial_GP_repo   # any module with
                                                                  # the reference's utils.py surface

The second form is how the whole-fit wall time of the reference's CPU path was taken in the build
container for DESIGN.md (BASELINE config[0]); nothing in this script is specific to either module
-- that is the drop-in boundary."""
import argparse
import contextlib
import io
import os
import sys
import time
import warnings

import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--d", type=int, default=64, help="pixels; a square grid")
ap.add_argument("--ntilde", type=int, default=None, help="inducing images (the first ntilde of the training set); default: all")
ap.add_argument("--maxiter", type=int, default=4)
ap.add_argument("--nestep", type=int, default=2)
ap.add_argument("--nmstep", type=int, default=6)
ap.add_argument("--nfstep", type=int, default=4)
ap.add_argument("--tol", type=float, default=1e-4, help="EIGVAL_TOL (1e-4 = the reference's default, truncating)")
ap.add_argument("--utils", default=None, help="directory holding a utils.py with the reference's surface")
ap.add_argument("--verbose", action="store_true")
ap.add_argument("--repeat", type=int, default=None,
                help="run the fit this many times from the same start; the first run carries the one-time costs of the "
                     "process (torch's optimizer machinery import ~0.7 s, library load, workspace allocation)")
args = ap.parse_args()
if args.repeat is None:
    args.repeat = 1 if args.utils else 2     # default: steady-state figure for this library, one run of a CPU module

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if args.utils:
    sys.path.insert(0, args.utils)
    sys.dont_write_bytecode = True
    os.environ.setdefault("MPLBACKEND", "Agg")
    with contextlib.redirect_stdout(io.StringIO()):
        import utils as gp
    backend = f"utils.py from {args.utils}"
else:
    from gaussian_processes_amd import utils as gp
    backend = "gaussian_processes_amd.utils (MI355X)"
from gaussian_processes_amd import synthetic as syn  # noqa: E402

n_px = int(round(args.d ** 0.5))
assert n_px * n_px == args.d, "--d must be a square number of pixels"
dev = torch.device("cuda") if (not args.utils and torch.cuda.is_available()) else torch.device("cpu")
X = torch.from_numpy(syn.stimuli(args.n, args.d)).to(dev)
r = torch.from_numpy(syn.cell_inputs(args.n, 0)[0]).to(dev)
rng = np.random.default_rng(7)
X_test = torch.from_numpy(rng.standard_normal((30, n_px, n_px, 1))).to(dev)   # images, as the notebook passes them
R_test = torch.from_numpy(rng.poisson(0.7, (10, 30, 1)).astype(np.float64)).to(dev)

lower, upper = syn.limits()


def fresh_start():
    theta = {k: torch.tensor(float(v), dtype=torch.float64, requires_grad=True) for k, v in syn.theta0().items()}
    ntilde = args.ntilde or args.n
    fit_parameters = {"ntilde": ntilde, "maxiter": args.maxiter, "nEstep": args.nestep, "nMstep": args.nmstep,
                      "nFparamstep": args.nfstep, "kernfun": "acosker", "cellid": 0, "n_px_side": n_px,
                      "display_hyper": False}
    return {"fit_parameters": fit_parameters, "xtilde": X if ntilde == args.n else X[:ntilde].clone(), "hyperparams_tuple": (theta, lower, upper),
            "f_params": {"logA": torch.tensor(syn.F_PARAMS["logA"], dtype=torch.float64, requires_grad=True),
                         "lambda0": torch.tensor(syn.F_PARAMS["lambda0"], dtype=torch.float64)}}


gp.EIGVAL_TOL = args.tol
sink = contextlib.nullcontext() if args.verbose else contextlib.redirect_stdout(io.StringIO())
fit_times = []
with sink, warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for rep in range(max(1, args.repeat)):
        t0 = time.time()
        fit_model, err_dict = gp.varGP(X, r, **fresh_start())
        if dev.type == "cuda":
            torch.cuda.synchronize()
        fit_times.append(time.time() - t0)
        if err_dict["is_error"]:
            break
    t_fit = fit_times[-1]
    t0 = time.time()
    R_test_cell, R_pred_cell, r2, sigma_r2 = gp.test(X_test, R_test, X_train=X, at_iteration=None, **fit_model)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    t_test = time.time() - t0
if err_dict["is_error"]:
    raise err_dict["error"]
lm = fit_model["values_track"]["loss_track"]["logmarginal"]
print(f"backend: {backend}")
print(f"N={args.n} ntilde={args.ntilde or args.n} d={args.d} EIGVAL_TOL={args.tol:g}: kept {fit_model['B'].shape[1]} of {args.ntilde or args.n} eigen-directions"
      f" (basis route {fit_model.get('basis_route', 'n/a')})")
print(f"varGP: {t_fit:.2f} s  ({args.maxiter} iterations x [{args.nestep} E, {args.nfstep} f-param, {args.nmstep} M]"
      + (f"; first run in this process {fit_times[0]:.2f} s" if len(fit_times) > 1 else "") + f");  test(): {t_test:.3f} s")
print("logmarginal per iteration:", " ".join(f"{float(v):.4f}" for v in lm))
print("final theta:", {k: round(float(v), 5) for k, v in fit_model["hyperparams_tuple"][0].items()})
print(f"predicted rates of the first 5 test images: {[round(float(v), 5) for v in R_pred_cell[:5]]}")
