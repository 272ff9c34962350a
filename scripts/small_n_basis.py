"""rocSOLVER's eigh against the two eigh-free routes for the kept eigenspace of small inducing sets
(profiles/r04_small_n_basis.log): `python scripts/small_n_basis.py`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_processes_amd import utils as gp, synthetic as syn, eigtop
dev = torch.device("cuda:0")
lower, upper = syn.limits()
def kt(NT, scale=1.0):
    X = torch.from_numpy(syn.stimuli(NT, 256)).to(dev)
    th = {k: torch.tensor(float(v), dtype=torch.float64) for k, v in syn.theta0().items()}
    th["sigma_0"] = th["sigma_0"] * scale
    C, mask = gp.localker(th, upper, lower, 16)
    Xm = X[:, mask].contiguous()
    return gp.acosker(th, Xm, Xm, C=C)
def timeit(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3, out
for NT in (256, 384, 512, 640, 768, 1000, 1024, 1280, 1408, 1536, 1792, 2048, 2100, 2560):
    K = kt(NT)
    t_eigh, (w, U) = timeit(lambda: torch.linalg.eigh(K))
    keep = int((w > max(float(w[-1]) * 1e-4, 1e-4)).sum())
    line = f"ntilde {NT}: eigh {t_eigh:.1f} ms kept {keep}"
    t_d, out = timeit(lambda: eigtop.kept_eigenspace_dense(K, 1e-4, gp.matmul, gp.cholesky, gemm_into=gp.gemm_into))
    if out is None:
        line += f" | dense projector {t_d:.1f} ms -> declined"
    else:
        n = out[2]["n"]
        if n == keep and 0 < n < NT:
            P = U[:, -n:].T @ out[1]
            err = float((P.T @ P - torch.eye(n, device=dev, dtype=torch.float64)).abs().max())
        else:
            err = float("nan")
        line += f" | dense projector {t_d:.1f} ms (kept {n}, {out[2]['sign_iterations']} sign steps, subspace dist {err:.1e})"
    if NT >= 1408:
        t_c, o = timeit(lambda: eigtop.top_eigenpairs(K, 1e-4, gp.matmul, gp.cholesky, basis="subspace", gemm_into=gp.gemm_into))
        if o is not None and o[0] is None:
            K2 = kt(NT, 1.003)
            t_w, o2 = timeit(lambda: eigtop.top_eigenpairs(K2, 1e-4, gp.matmul, gp.cholesky, basis="subspace", gemm_into=gp.gemm_into, start=o[2]["state"]))
            line += f" | sweeps k {o[2]['k']}: cold {t_c:.1f} ms ({o[2]['sweeps']} sw), warm {t_w:.1f} ms ({o2[2]['sweeps'] if o2 is not None and o2[0] is None else 'x'} sw)"
            if out is not None:
                line += f", canonical bases differ by {float((o[1] - out[1]).abs().max()):.1e}"
    print(line, flush=True)
