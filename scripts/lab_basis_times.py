"""Warm-started basis build at the lab's inducing-set size (2100): time split with synchronised timers."""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_processes_amd import utils as gp, synthetic as syn, eigtop
NT = int(sys.argv[1]) if len(sys.argv) > 1 else 2100
dev = torch.device("cuda:0")
X = torch.from_numpy(syn.stimuli(NT, 256)).to(dev)
lower, upper = syn.limits()
def kt(scale):
    th = {k: torch.tensor(float(v), dtype=torch.float64) for k, v in syn.theta0().items()}
    th["sigma_0"] = th["sigma_0"] * scale
    th["-2log2beta"] = th["-2log2beta"] + (scale - 1.0)
    C, mask = gp.localker(th, upper, lower, 16)
    Xm = X[:, mask].contiguous()
    return gp.acosker(th, Xm, Xm, C=C)
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
gi = timed("gemm_into", gp.gemm_into)
K0 = kt(1.0)
out = eigtop.top_eigenpairs(K0, 1e-4, gp.matmul, gp.cholesky, basis="subspace", gemm_into=gp.gemm_into)
state = out[2]["state"]
for scale in (1.0, 1.002, 1.01, 1.05):
    K1 = kt(scale)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o = eigtop.top_eigenpairs(K1, 1e-4, gp.matmul, gp.cholesky, basis="subspace", gemm_into=gp.gemm_into, start=state)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    acc.clear()
    t_all = time.perf_counter()
    o = eigtop.top_eigenpairs(K1, 1e-4, mm, chol, basis="subspace", gemm_into=gi, start=state, log=print)
    torch.cuda.synchronize(); t_all = (time.perf_counter() - t_all) * 1e3
    info = {k: v for k, v in o[2].items() if not torch.is_tensor(v) and k != "state"}
    print(f"scale {scale}: warm {ms:.2f} ms (instrumented {t_all:.1f}); kept {o[1].shape[1]} info {info}")
    tot = 0
    for name, (cnt, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"    {name:58s} n {cnt:3d} total {t*1e3:7.2f} ms  avg {t/cnt*1e3:6.3f} ms"); tot += t
    print(f"    sum of timed primitives {tot*1e3:.2f} ms")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o = eigtop.top_eigenpairs(K1, 1e-4, gp.matmul, gp.cholesky, basis="subspace", gemm_into=gp.gemm_into)
    torch.cuda.synchronize(); print(f"    cold: {(time.perf_counter() - t0) * 1e3:.2f} ms sweeps {o[2].get('sweeps')}")
