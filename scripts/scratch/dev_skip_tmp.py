import os, sys, time, ctypes
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from gaussian_processes_amd import synthetic as syn, _lib
from gaussian_processes_amd.engine import GPFitEngine
import bench
N = int(sys.argv[1]); d = 256; reps = 8
dev = torch.device("cuda:0")
grid = syn.grid_for(d); lower, upper = syn.limits()
X = torch.from_numpy(syn.stimuli(N, d)).to(dev)
r_np, m_np = syn.cell_inputs(N); r, m = torch.from_numpy(r_np).to(dev), torch.from_numpy(m_np).to(dev)
eng = GPFitEngine(N, d)
V = bench.build_V(X, grid, syn.theta0(), dev)
th1 = syn.theta_eval(); logA, lam0 = syn.F_PARAMS["logA"], syn.F_PARAMS["lambda0"]
def step():
    t = eng.fit_eval_async(th1, lower, upper, grid, X, r, m, V, logA, lam0, want_vectors=False, _sync=True)
    return t["rc"]
for _ in range(3): rc = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): rc = step()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
eng.set_profile(2); acc = None
for _ in range(reps):
    step(); ph = eng.get_phases()
    acc = ph if acc is None else {k: acc[k] + ph[k] for k in ph}
print(f"SKIP_TMP={os.environ.get('GPFIT_DEV_SKIP_TMP', '0')} N={N}: {ms:.3f} ms/fit rc {rc} phases", {k: round(v / reps, 3) for k, v in acc.items()})
