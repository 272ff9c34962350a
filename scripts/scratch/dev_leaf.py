"""Time the Cholesky leaf alone via a potrf on a 128-padded problem (N=128 -> one leaf) and via
gpfit_fit_eval profile at N=8192 for phase ablations (GPFIT_LEAF_DBG)."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaussian_processes_amd import synthetic as syn
from gaussian_processes_amd.engine import GPFitEngine
dev = torch.device("cuda:0")
N, d = 8192, 256
grid = syn.grid_for(d); lower, upper = syn.limits()
X = torch.from_numpy(syn.stimuli(N, d)).to(dev)
r_np, m_np = syn.cell_inputs(N)
r, m = torch.from_numpy(r_np).to(dev), torch.from_numpy(m_np).to(dev)
eng = GPFitEngine(N, d)
V = torch.eye(N, dtype=torch.float64, device=dev) * 2.0 + 0.5
eng.set_profile(True)
for _ in range(2):
    try:
        eng.fit_eval(syn.theta_eval(), lower, upper, grid, X, r, m, V, syn.F_PARAMS["logA"], syn.F_PARAMS["lambda0"], want_grad=False, want_vectors=False)
    except Exception as e:
        pass
p = eng.get_profile()
print(f"dbg={os.environ.get('GPFIT_LEAF_DBG','0')}: leaf {p['leaf_ms']/max(1,p['leaf_launches'])*1e3:.1f} us avg over {p['leaf_launches']} launches")
