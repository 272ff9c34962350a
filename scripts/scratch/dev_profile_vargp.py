"""cProfile of the whole varGP fit through the drop-in (where does the host time go?)."""
import cProfile, contextlib, io, os, pstats, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaussian_processes_amd import utils as gp, synthetic as syn
N, d = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 256
X = torch.from_numpy(syn.stimuli(N, d)).cuda(); r = torch.from_numpy(syn.cell_inputs(N, 0)[0]).cuda()
lower, upper = syn.limits()
def run():
    theta = {k: torch.tensor(float(v), dtype=torch.float64, requires_grad=True) for k, v in syn.theta0().items()}
    fp = {"ntilde": N, "maxiter": 4, "nEstep": 2, "nMstep": 6, "nFparamstep": 4, "kernfun": "acosker", "cellid": 0, "n_px_side": 16, "display_hyper": False}
    init = {"fit_parameters": fp, "xtilde": X, "hyperparams_tuple": (theta, lower, upper),
            "f_params": {"logA": torch.tensor(syn.F_PARAMS["logA"], dtype=torch.float64, requires_grad=True), "lambda0": torch.tensor(syn.F_PARAMS["lambda0"], dtype=torch.float64)}}
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fit, err = gp.varGP(X, r, **init)
    torch.cuda.synchronize()
    assert not err["is_error"], err
run()
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
