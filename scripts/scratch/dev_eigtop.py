"""Block subspace iteration (gaussian_processes_amd/eigtop.py) against torch.linalg.eigh on the bench kernel matrix."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gaussian_processes_amd import utils as gp, synthetic as syn, eigtop
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = 256
dev = torch.device("cuda:0")
X = torch.from_numpy(syn.stimuli(N, d)).to(dev)
K = 2.0 * bench.build_V(X, syn.grid_for(d), syn.theta0(), dev)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter(); w, U = torch.linalg.eigh(K, UPLO='L'); torch.cuda.synchronize(); t_eigh = time.perf_counter() - t0
tau = max(float(w[-1]) * 1e-4, 1e-4); keep = w > tau
print(f"N={N}: eigh {t_eigh*1e3:.1f} ms, kept {int(keep.sum())}")
for rep in range(2):
    t0 = time.perf_counter(); out = eigtop.top_eigenpairs(K, 1e-4, gp.matmul, gp.cholesky, log=print if rep == 0 else None); torch.cuda.synchronize(); t_sub = time.perf_counter() - t0
if out is None:
    print("fallback"); sys.exit(0)
vals, vecs, info = out
print(f"subspace: {t_sub*1e3:.1f} ms {info}")
Bref = U[:, keep]
print("n equal:", vals.shape[0] == int(keep.sum()), " eigenvalue rel err:", float(((vals - w[keep]).abs() / w[keep]).max()))
P = Bref.T @ vecs
print("subspace distance ||I - P^T P||:", float((torch.eye(P.shape[1], device=dev, dtype=torch.float64) - P.T @ P).abs().max()),
      " orthonormality:", float((vecs.T @ vecs - torch.eye(vecs.shape[1], device=dev, dtype=torch.float64)).abs().max()))
# block size / first-sweep sweep (which start needs the fewest products and Rayleigh-Ritz solves?)
for k0 in (640, 768, 896, 1024):
    for fs in (8, 12, 16):
        eigtop.top_eigenpairs(K, 1e-4, gp.matmul, gp.cholesky, k0=k0, first_sweeps=fs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o = eigtop.top_eigenpairs(K, 1e-4, gp.matmul, gp.cholesky, k0=k0, first_sweeps=fs)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"k0 {k0} first_sweeps {fs}: {dt*1e3:.1f} ms", None if o is None else {k: o[2][k] for k in ("k", "sweeps", "rr", "products", "grown", "n")})
for k in (512, 768, 1024):
    S = torch.randn(k, k, device=dev, dtype=torch.float64); S = S + S.T
    torch.linalg.eigh(S); torch.cuda.synchronize(); t0 = time.perf_counter(); torch.linalg.eigh(S); torch.cuda.synchronize()
    print(f"torch.linalg.eigh {k} x {k}: {(time.perf_counter()-t0)*1e3:.1f} ms")
