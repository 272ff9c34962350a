"""A/B of the factorisation schedules: python scripts/scratch/dev_lockstep.py [N] [d] [reps]
Prints ms/fit, phases and the hex of every output scalar (run under GPFIT_LOCKSTEP=0/1, GPFIT_NO_BATCH=1 and diff)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaussian_processes_amd import synthetic as syn
from gaussian_processes_amd.engine import GPFitEngine
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
grid = syn.grid_for(d); lower, upper = syn.limits()
X = torch.from_numpy(syn.stimuli(N, d)).to(dev)
r_np, m_np = syn.cell_inputs(N); r, m = torch.from_numpy(r_np).to(dev), torch.from_numpy(m_np).to(dev)
eng = GPFitEngine(N, d)
V = bench.build_V(X, grid, syn.theta0(), dev)
th1 = syn.theta_eval(); logA, lam0 = syn.F_PARAMS["logA"], syn.F_PARAMS["lambda0"]
def step(**kw): return eng.fit_eval(th1, lower, upper, grid, X, r, m, V, logA, lam0, want_vectors=False, **kw)
for _ in range(3): res = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): res = step()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
tag = f"LOCKSTEP={os.environ.get('GPFIT_LOCKSTEP', 'default')} NO_BATCH={os.environ.get('GPFIT_NO_BATCH', '-')}"
print(f"{tag} N={N} d={d}: {ms:.3f} ms/fit (host enqueue {eng.last_enqueue_ms():.2f} ms)")
eng.set_profile(2); acc = None
for _ in range(reps):
    step(); ph = eng.get_phases()
    acc = ph if acc is None else {k: acc[k] + ph[k] for k in ph}
eng.set_profile(0)
print("  phases", {k: round(v / reps, 3) for k, v in acc.items()})
eng.set_profile(1); step(); p = eng.get_profile(); eng.set_profile(0)
print(f"  profile: gemm128 {p['gemm_ms']:.2f} ms / {p['gemm_launches']}; leaf {p['leaf_ms']:.2f} ms / {p['leaf_launches']}; small gemm {p['small_gemm_ms']:.2f} ms / {p['small_gemm_launches']}")
print("  hex", float(res["loss"]).hex(), " ".join(float(v).hex() for v in res["grad"].values()), float(res["logdet_V"]).hex(), float(res["tr_KinvV"]).hex())
if len(sys.argv) > 4:
    res32 = step(grad_precision="f32")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): res32 = step(grad_precision="f32")
    torch.cuda.synchronize(); print(f"  mixed: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms/fit  hex {float(res32['loss']).hex()} {' '.join(float(v).hex() for v in res32['grad'].values())}")
