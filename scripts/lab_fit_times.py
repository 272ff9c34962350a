"""Phase times of a lab-shaped fit (3160 training / 2100 inducing images, maxiter 30, nEstep 10, nMstep 10)."""
import contextlib, io, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussian_processes_amd import utils as gp, synthetic as syn
N, NT, d = 3160, 2100, 256
n_px = 16
dev = torch.device("cuda")
X = torch.from_numpy(syn.stimuli(N, d)).to(dev)
r = torch.from_numpy(syn.cell_inputs(N, 0)[0]).to(dev)
lower, upper = syn.limits()
def start():
    theta = {k: torch.tensor(float(v), dtype=torch.float64, requires_grad=True) for k, v in syn.theta0().items()}
    fp = {"ntilde": NT, "maxiter": 30, "nEstep": 10, "nMstep": 10, "nFparamstep": 4, "kernfun": "acosker", "cellid": 0,
          "n_px_side": n_px, "display_hyper": False}
    return {"fit_parameters": fp, "xtilde": X[:NT].clone(), "hyperparams_tuple": (theta, lower, upper),
            "f_params": {"logA": torch.tensor(syn.F_PARAMS["logA"], dtype=torch.float64, requires_grad=True),
                         "lambda0": torch.tensor(syn.F_PARAMS["lambda0"], dtype=torch.float64)}}
for rep in range(3):
    buf = io.StringIO()
    t0 = time.time()
    with contextlib.redirect_stdout(buf):
        fm, err = gp.varGP(X, r, **start())
    torch.cuda.synchronize()
    wall = time.time() - t0
    tail = [l for l in buf.getvalue().splitlines() if l.startswith("Time spent")]
    print(f"rep {rep}: {wall:.3f} s", "|", " ".join(l.replace("Time spent ", "").strip() for l in tail), flush=True)
if len(sys.argv) > 1:
    import cProfile, pstats
    pr = cProfile.Profile()
    with contextlib.redirect_stdout(io.StringIO()):
        pr.enable(); fm, err = gp.varGP(X, r, **start()); torch.cuda.synchronize(); pr.disable()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(35)
