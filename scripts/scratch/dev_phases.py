"""Phase timing of the headline unit (no profiler): python scripts/scratch/dev_phases.py [N] [d] [reps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaussian_processes_amd import synthetic as syn
from gaussian_processes_amd.engine import GPFitEngine
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
grid = syn.grid_for(d); lower, upper = syn.limits()
X = torch.from_numpy(syn.stimuli(N, d)).to(dev)
r_np, m_np = syn.cell_inputs(N); r, m = torch.from_numpy(r_np).to(dev), torch.from_numpy(m_np).to(dev)
eng = GPFitEngine(N, d)
V = bench.build_V(X, grid, syn.theta0(), dev)
th1 = syn.theta_eval(); logA, lam0 = syn.F_PARAMS["logA"], syn.F_PARAMS["lambda0"]
def step(**kw): return eng.fit_eval(th1, lower, upper, grid, X, r, m, V, logA, lam0, want_vectors=False, **kw)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): step()
torch.cuda.synchronize(); print(f"N={N} d={d}: {(time.perf_counter()-t0)/reps*1e3:.3f} ms/fit (host enqueue {eng.last_enqueue_ms():.2f} ms)")
eng.set_profile(2)
acc = None
for _ in range(reps):
    step(); ph = eng.get_phases()
    acc = ph if acc is None else {k: acc[k] + ph[k] for k in ph}
eng.set_profile(0)
print({k: round(v / reps, 3) for k, v in acc.items()})
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): step(reuse_V=True)
torch.cuda.synchronize(); print(f"with the V factor reused: {(time.perf_counter()-t0)/reps*1e3:.3f} ms/fit")
for reuse in (False, True):
    eng.set_profile(1); step(reuse_V=reuse); p = eng.get_profile(); eng.set_profile(0)
    print(f"profile reuse_V={reuse}: gemm128 {p['gemm_ms']:.2f} ms / {p['gemm_launches']} launches; leaf {p['leaf_ms']:.2f} ms / {p['leaf_launches']} "
          f"({p['leaf_ms']/max(1,p['leaf_launches'])*1e3:.1f} us); small gemm {p['small_gemm_ms']:.2f} ms / {p['small_gemm_launches']} ({p['small_gemm_ms']/max(1,p['small_gemm_launches'])*1e3:.1f} us)")
