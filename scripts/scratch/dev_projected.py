"""Timing of the truncated-rank closure: fused entry vs step-by-step formulation (N=4096, d=256)."""
import os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaussian_processes_amd import utils as gp, synthetic as syn
KEYS = syn.THETA_KEYS; LOWER, UPPER = syn.limits()
def tth(v): return {k: torch.tensor(float(x), dtype=torch.float64, requires_grad=True) for k, x in zip(KEYS, v)}
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = 256
th = tth([syn.theta_eval()[k] for k in KEYS])
X = torch.from_numpy(syn.stimuli(N, d)).cuda(); r = torch.from_numpy(syn.cell_inputs(N)[0]).cuda(); m = torch.from_numpy(syn.cell_inputs(N)[1]).cuda()
C, mask = gp.localker(th, UPPER, LOWER, 16)
Kt = gp.acosker(th, X, X, C=C)
ev, evec = torch.linalg.eigh(Kt); keep = ev > max(float(ev.max()) * 1e-4, 1e-4)
B = evec[:, keep].contiguous(); print("kept", B.shape[1], "of", N)
m_b = gp.matmul(B, m, transA=True); V_b = gp.matmul(B, gp.matmul(0.5 * Kt, B), transA=True); V_b = (V_b + V_b.T) / 2
fp = {"logA": torch.tensor(syn.F_PARAMS["logA"], dtype=torch.float64), "lambda0": torch.tensor(syn.F_PARAMS["lambda0"], dtype=torch.float64)}
del Kt, evec
for name, fn in (("fused", gp._closure_projected), ("steps", gp._closure_projected_steps)):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(3): out = fn(th, (LOWER, UPPER), 16, X, r, B, m_b, V_b, fp)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): out = fn(th, (LOWER, UPPER), 16, X, r, B, m_b, V_b, fp)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"{name}: {dt*1e3:.2f} ms per closure, loss {out[0]:.9f} grad {[round(out[1][k], 6) for k in KEYS]}")
# sparse closure: n_t = N, n_tilde = N/2
nt_ = N // 2
xt = X[:nt_].contiguous()
Kt = gp.acosker(th, xt, xt, C=C)
ev, evec = torch.linalg.eigh(Kt); keep = ev > max(float(ev.max()) * 1e-4, 1e-4)
B = evec[:, keep].contiguous(); print("sparse: n_t", N, "n_tilde", nt_, "kept", B.shape[1])
m_b = gp.matmul(B, m[:nt_].contiguous(), transA=True); V_b = torch.diag(ev[keep]) * 0.5
del Kt, evec
for name, fn in (("fused", gp._closure_sparse), ("steps", gp._closure_sparse_steps)):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(3): out = fn(th, (LOWER, UPPER), 16, X, xt, r, B, m_b, V_b, fp)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): out = fn(th, (LOWER, UPPER), 16, X, xt, r, B, m_b, V_b, fp)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"sparse {name}: {dt*1e3:.2f} ms per closure, loss {out[0]:.9f} grad {[round(out[1][k], 6) for k in KEYS]}")
