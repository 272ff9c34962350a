"""Register-resident leaf against torch (factor and inverse of 128-blocks and the recursion above it),
and leaf timing through the profile counters.  GPFIT_LEAF_LDS=1 selects the old LDS-resident leaf."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaussian_processes_amd import utils as gp, synthetic as syn
from gaussian_processes_amd.engine import GPFitEngine
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for n in (16, 100, 128, 129, 256, 300, 1024, 2000):
    M = torch.randn(n, n, dtype=torch.float64, generator=g)
    S = (M @ M.T + n * torch.eye(n, dtype=torch.float64)).to(dev)
    L, Li, logdet, info = gp.cholesky(S, want_inverse=True)
    Lref = torch.linalg.cholesky(S)
    eL = float((L - Lref).abs().max() / Lref.abs().max())
    eI = float((Li @ Lref - torch.eye(n, dtype=torch.float64, device=dev)).abs().max())
    up = float(torch.triu(L, 1).abs().max()) if n > 1 else 0.0
    upi = float(torch.triu(Li, 1).abs().max()) if n > 1 else 0.0
    print(f"n={n:5d} info={info} relerr L {eL:.2e}  |Li L - I| {eI:.2e}  upper(L) {up:.1e} upper(Li) {upi:.1e} logdet err {abs(logdet-float(torch.logdet(S))):.2e}")
# non-SPD: info
S = torch.eye(200, dtype=torch.float64, device=dev); S[150, 150] = -1.0
print("info for a negative pivot at 151:", gp.cholesky(S)[3])
# timing in the fit
N, d = 8192, 256
grid = syn.grid_for(d); lower, upper = syn.limits()
X = torch.from_numpy(syn.stimuli(N, d)).to(dev)
r_np, m_np = syn.cell_inputs(N); r, m = torch.from_numpy(r_np).to(dev), torch.from_numpy(m_np).to(dev)
eng = GPFitEngine(N, d)
import bench
V = bench.build_V(X, grid, syn.theta0(), dev)
th1 = syn.theta_eval()
def step(): return eng.fit_eval(th1, lower, upper, grid, X, r, m, V, syn.F_PARAMS["logA"], syn.F_PARAMS["lambda0"], want_vectors=False)
for _ in range(2): res = step()
print("loss", res["loss"])
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); print(f"{(time.perf_counter()-t0)/5*1e3:.3f} ms/fit")
eng.set_profile(1); step(); p = eng.get_profile(); eng.set_profile(0)
print(f"leaf {p['leaf_ms']/max(1,p['leaf_launches'])*1e3:.1f} us avg over {p['leaf_launches']} launches")
eng.set_profile(2); step(); print(eng.get_phases())
