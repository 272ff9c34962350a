"""Whole default-tolerance fits (6 EM iterations x [4 E, 4 f-param, 6 M]) over a grid of training / inducing set sizes: wall time of
the second fit of each shape, kept count, the basis routes taken (profiles/r04_shape_sweep.log)."""
import contextlib, io, sys, time, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussian_processes_amd import utils as gp, synthetic as syn
dev = torch.device("cuda")
lower, upper = syn.limits()
d, n_px = 256, 16
def fit(N, NT, maxiter=6):
    X = torch.from_numpy(syn.stimuli(N, d)).to(dev)
    r = torch.from_numpy(syn.cell_inputs(N, 0)[0]).to(dev)
    theta = {k: torch.tensor(float(v), dtype=torch.float64, requires_grad=True) for k, v in syn.theta0().items()}
    fp = {"ntilde": NT, "maxiter": maxiter, "nEstep": 4, "nMstep": 6, "nFparamstep": 4, "kernfun": "acosker", "cellid": 0, "n_px_side": n_px, "display_hyper": False}
    args = {"fit_parameters": fp, "xtilde": X if NT == N else X[:NT].clone(), "hyperparams_tuple": (theta, lower, upper),
            "f_params": {"logA": torch.tensor(syn.F_PARAMS["logA"], dtype=torch.float64, requires_grad=True), "lambda0": torch.tensor(syn.F_PARAMS["lambda0"], dtype=torch.float64)}}
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.cuda.synchronize(); t0 = time.time()
        fm, err = gp.varGP(X, r, **args)
        torch.cuda.synchronize(); dt = time.time() - t0
    return fm, err, dt
fit(600, 300)
for N in (600, 1000, 1500, 2000, 3160):
    for NT in (300, 500, 800, 1200, 1600, 2100, 3160):
        if NT > N:
            continue
        fm, err, dt = fit(N, NT)
        fm, err, dt = fit(N, NT)
        routes = fm["values_track"]["variation_par_track"]["basis_route"] if not err["is_error"] else ()
        lm = fm["values_track"]["loss_track"]["logmarginal"]
        print(f"n_t {N:5d} n_tilde {NT:5d}: {dt*1e3:7.1f} ms  kept {fm['B'].shape[1]:4d}  routes {sorted(set(routes))}  error {err['is_error']}  logmarginal {float(lm[0]):.3f} -> {float(lm[-1]):.3f}", flush=True)
