"""Where a whole EM fit (varGP through the drop-in module) spends its time, at the reference's default tolerance.

    python scripts/whole_fit_breakdown.py [N] [d] [out.json] [ntilde] [maxiter,nEstep,nMstep,nFparamstep]

Two runs of the same fit from the same start: one untouched (the wall time that counts), one with every entry
point of the C ABI and the few torch routines the host side uses (eigh, L-BFGS step) wrapped in synchronised
timers -- device time per entry point (exclusive: nested calls are charged to the innermost), call counts, and
what is left for the Python host loop.  The instrumented run is slower than the untouched one by the overlap
the synchronisation removes; the JSON reports both."""
import contextlib
import io
import json
import os
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import _lib, synthetic as syn, utils as gp  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
out_path = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else None
NTILDE = int(sys.argv[4]) if len(sys.argv) > 4 else N          # < N: the sparse regime (the lab's runs: n_t = 3160, ntilde <= 2100)
n_px = int(round(d ** 0.5))
dev = torch.device("cuda")
X = torch.from_numpy(syn.stimuli(N, d)).to(dev)
r = torch.from_numpy(syn.cell_inputs(N, 0)[0]).to(dev)
lower, upper = syn.limits()
SETTINGS = {"maxiter": 4, "nEstep": 2, "nMstep": 6, "nFparamstep": 4}
if len(sys.argv) > 5:
    SETTINGS = dict(zip(("maxiter", "nEstep", "nMstep", "nFparamstep"), (int(v) for v in sys.argv[5].split(","))))


def start():
    theta = {k: torch.tensor(float(v), dtype=torch.float64, requires_grad=True) for k, v in syn.theta0().items()}
    fp = dict(SETTINGS, ntilde=NTILDE, kernfun="acosker", cellid=0, n_px_side=n_px, display_hyper=False)
    return {"fit_parameters": fp, "xtilde": X if NTILDE == N else X[:NTILDE].clone(), "hyperparams_tuple": (theta, lower, upper),
            "f_params": {"logA": torch.tensor(syn.F_PARAMS["logA"], dtype=torch.float64, requires_grad=True),
                         "lambda0": torch.tensor(syn.F_PARAMS["lambda0"], dtype=torch.float64)}}


def fit():
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model, err = gp.varGP(X, r, **start())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert not err["is_error"], err
    return model, dt


fit()                                   # process warm-up (library load, workspace, torch's optimizer import)
model, wall = fit()
tracks = model["values_track"]["loss_track"]["logmarginal"].tolist()

# ---- instrumented run
acc = {}          # name -> [calls, seconds]
stack = []


def timed(name, fn):
    def wrapper(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stack.append(0.0)
        try:
            return fn(*a, **k)
        finally:
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            inner = stack.pop()
            rec = acc.setdefault(name, [0, 0.0])
            rec[0] += 1
            rec[1] += dt - inner
            if stack:
                stack[-1] += dt
    return wrapper


lib = _lib.load()
originals = {}
for name in _lib.exported_symbols():
    if name in ("gpfit_last_error", "gpfit_version", "gpfit_last_enqueue_ms"):
        continue
    originals[name] = getattr(lib, name)
    setattr(lib, name, timed(name, originals[name]))
eigh0, step0 = torch.linalg.eigh, torch.optim.LBFGS.step
torch.linalg.eigh = timed("torch.linalg.eigh", eigh0)
basis0, eig0 = gp._stabilised_basis, gp.eigtop.top_eigenpairs
gp._stabilised_basis = timed("host: _stabilised_basis (rank decision, basis assembly)", basis0)
gp.eigtop.top_eigenpairs = timed("host: eigtop.top_eigenpairs (subspace iteration driver)", eig0)
torch.optim.LBFGS.step = timed("host: torch.optim.LBFGS.step (both optimisers: line search, history, closures' Python)", step0)
model2, wall_instr = fit()
for name, fn in originals.items():
    setattr(lib, name, fn)
torch.linalg.eigh, torch.optim.LBFGS.step = eigh0, step0
gp._stabilised_basis, gp.eigtop.top_eigenpairs = basis0, eig0

rows = sorted(((n, c, s) for n, (c, s) in acc.items()), key=lambda t: -t[2])
accounted = sum(s for _, _, s in rows)
report = {
    "what": f"varGP at N={N} ntilde={NTILDE} d={d}, EIGVAL_TOL={gp.EIGVAL_TOL} (the reference's default), {SETTINGS}, steady state (second fit of the process)",
    "wall_s": round(wall, 4), "wall_instrumented_s": round(wall_instr, 4),
    "basis_route_per_iteration": list(model["values_track"]["variation_par_track"]["basis_route"]),
    "n_kept": int(model["B"].shape[1]), "logmarginal_track": tracks,
    "exclusive_seconds": [{"where": n, "calls": c, "s": round(s, 4), "share_of_instrumented": round(s / wall_instr, 3)} for n, c, s in rows],
    "unaccounted_host_s": round(wall_instr - accounted, 4),
}
txt = json.dumps(report, indent=1)
print(txt)
if out_path:
    open(out_path, "w").write(txt + "\n")
