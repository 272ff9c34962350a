"""Time split of eigtop.top_eigenpairs on the bench kernel matrix (the primitives wrapped in synchronised timers):
`python scripts/eigtop_parts.py N [k0]` -- plain sweeps against the Chebyshev-shifted ones (profiles/r04_eigtop.log)."""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_processes_amd import utils as gp, synthetic as syn, eigtop
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
X = torch.from_numpy(syn.stimuli(N, 256)).to(dev)
K = 2.0 * bench.build_V(X, syn.grid_for(256), syn.theta0(), dev)
w, U = torch.linalg.eigh(K)
keep = w > max(float(w[-1]) * 1e-4, 1e-4)
Bref = U[:, keep]
acc = collections.defaultdict(lambda: [0, 0.0])
def timed(name, fn):
    def wrapped(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(*a, **k); torch.cuda.synchronize()
        acc[name][0] += 1; acc[name][1] += time.perf_counter() - t0; return r
    return wrapped
def mm(A, B, **k):
    key = f"matmul {tuple(A.shape)}{'T' if k.get('transA') else ''} x {tuple(B.shape)}{'T' if k.get('transB') else ''}"
    return timed(key, gp.matmul)(A, B, **k)
chol = timed("cholesky+inverse", gp.cholesky)
orig_eigh = torch.linalg.eigh
for accel, k0, basis in ((False, None, "eigenvectors"), (True, None, "eigenvectors"), (True, None, "subspace"), (True, 896, "subspace")):
    torch.linalg.eigh = orig_eigh
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = eigtop.top_eigenpairs(K, 1e-4, gp.matmul, gp.cholesky, k0=k0, accelerate=accel, basis=basis)
        torch.cuda.synchronize(); plain_ms = (time.perf_counter() - t0) * 1e3
    torch.linalg.eigh = timed("eigh (k x k)", orig_eigh)
    acc.clear()
    out = eigtop.top_eigenpairs(K, 1e-4, mm, chol, k0=k0, accelerate=accel, log=print, basis=basis)
    vals, vecs, info = out
    P = Bref.T @ vecs
    err = float((P.T @ P - torch.eye(P.shape[1], device=dev, dtype=torch.float64)).abs().max())
    if vals is None:
        vals = torch.linalg.eigvalsh(info["K_tilde_b"])
    small = {k: v for k, v in info.items() if not torch.is_tensor(v)}
    print(f"N={N} accelerate={accel} k0={k0} basis={basis}: {plain_ms:.1f} ms; kept {vecs.shape[1]} (eigh: {int(keep.sum())}), eigenvalue rel err "
          f"{float(((vals - w[keep]).abs() / w[keep]).max()):.1e}, subspace distance {err:.1e}, info {small}")
    for name, (cnt, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"    {name:58s} n {cnt:3d} total {t*1e3:7.1f} ms  avg {t/cnt*1e3:6.2f} ms")
torch.linalg.eigh = orig_eigh
