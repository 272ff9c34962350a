import os, sys, time, io, contextlib, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gaussian_processes_amd import utils as gp, synthetic as syn
N, d, n_px = 8192, 256, 16
dev = torch.device("cuda")
X = torch.from_numpy(syn.stimuli(N, d)).to(dev); r = torch.from_numpy(syn.cell_inputs(N, 0)[0]).to(dev)
rng = np.random.default_rng(7)
X_test = torch.from_numpy(rng.standard_normal((30, n_px, n_px, 1))).to(dev)
R_test = torch.from_numpy(rng.poisson(0.7, (10, 30, 1)).astype(np.float64)).to(dev)
lower, upper = syn.limits()
def start():
    theta = {k: torch.tensor(float(v), dtype=torch.float64, requires_grad=True) for k, v in syn.theta0().items()}
    fp = {"ntilde": N, "maxiter": 2, "nEstep": 2, "nMstep": 3, "nFparamstep": 3, "kernfun": "acosker", "cellid": 0, "n_px_side": n_px, "display_hyper": False}
    return {"fit_parameters": fp, "xtilde": X, "hyperparams_tuple": (theta, lower, upper),
            "f_params": {"logA": torch.tensor(syn.F_PARAMS["logA"], dtype=torch.float64, requires_grad=True), "lambda0": torch.tensor(syn.F_PARAMS["lambda0"], dtype=torch.float64)}}
with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
    warnings.simplefilter("ignore")
    fit, err = gp.varGP(X, r, **start())
    ts = []
    for i in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        gp.test(X_test, R_test, X_train=X, at_iteration=None, **fit)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("test() times:", [f"{t*1e3:.1f} ms" for t in ts], "B", tuple(fit["B"].shape), fit["B"].is_contiguous(), fit["K_tilde_inv_b"].shape)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
with contextlib.redirect_stdout(io.StringIO()):
    gp.test(X_test, R_test, X_train=X, at_iteration=None, **fit); torch.cuda.synchronize()
pr.disable(); s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14); print("\n".join(l[:140] for l in s.getvalue().splitlines()[:30]))
