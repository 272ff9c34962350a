"""Side measurements for DESIGN.md: (1) is an fp32 Cholesky of K~ viable at the headline size?
(2) the truncated-rank (default EIGVAL_TOL) closure on the GPU primitives; (3) fused E-step."""
import os, sys, time, warnings, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import utils as gp, synthetic as syn, _lib
dev = torch.device("cuda:0")
lower, upper = syn.limits()
tth = lambda th: {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in th.items()}

def K_of(N, d, th):
    grid = syn.grid_for(d)
    X = torch.from_numpy(syn.stimuli(N, d)).to(dev)
    C, mask = gp.localker(tth(th), upper, lower, grid)
    return X, gp.acosker(tth(th), X, X, C=C)

# (1) fp32 feasibility
for N in (2048, 4096, 8192):
    X, K = K_of(N, 256, syn.theta_eval())
    ev = torch.linalg.eigvalsh(K)
    cond = float(ev.max() / ev.min())
    L64 = torch.linalg.cholesky(K)
    ld64 = float(2 * torch.log(torch.diagonal(L64)).sum())
    K32 = K.to(torch.float32)
    L32, info = torch.linalg.cholesky_ex(K32)
    ld32 = float(2 * torch.log(torch.diagonal(L32).double()).sum()) if int(info) == 0 else float("nan")
    print(f"fp32 check N={N}: cond(K~)={cond:.3e} lambda_min={float(ev.min()):.3e} fp32 potrf info={int(info)} "
          f"logdet fp64={ld64:.6f} fp32={ld32:.6f} rel err={abs(ld32-ld64)/abs(ld64):.2e}", flush=True)
    del K, K32, L64, L32

# (2) truncated-rank closure at N=4096, d=256, default tolerance
N, d = 4096, 256
grid = syn.grid_for(d)
X, K0 = K_of(N, d, syn.theta0())
r_np, m_np = syn.cell_inputs(N)
r, m = torch.from_numpy(r_np).to(dev), torch.from_numpy(m_np).to(dev)
gp.EIGVAL_TOL = 1e-4
ev, evec, keep = gp._eigen_stabilise(K0)
B = evec[:, keep].contiguous()
print(f"truncated regime N={N} d={d}: kept {B.shape[1]} of {N}", flush=True)
m_b = gp.matmul(B, m, transA=True)
V_b = gp.matmul(B, gp.matmul(0.5 * K0, B), transA=True)
V_b = (V_b + V_b.T) / 2
fp = {"logA": torch.tensor(syn.F_PARAMS["logA"]), "lambda0": torch.tensor(syn.F_PARAMS["lambda0"])}
th1 = tth(syn.theta_eval())
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        loss, grad = gp._closure_general(th1, (lower, upper), grid, X, X, r, B, m_b, V_b, fp, N, N)
        torch.cuda.synchronize(); print(f"  general closure: {time.time()-t0:.3f} s  loss {loss:.6f}", flush=True)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        loss2, grad2 = gp._closure_projected(th1, (lower, upper), grid, X, r, B, m_b, V_b, fp)
        torch.cuda.synchronize(); print(f"  projected adjoint closure: {time.time()-t0:.4f} s  loss {loss2:.6f}  max grad dev "
                                        f"{max(abs(grad[k]-grad2[k]) for k in grad)/max(abs(v) for v in grad.values()):.2e}", flush=True)

# (3) fused E-step at the headline size
N, d = 8192, 256
X, K = K_of(N, d, syn.theta_eval())
r_np, m_np = syn.cell_inputs(N)
r, m = torch.from_numpy(r_np).to(dev), torch.from_numpy(m_np).to(dev)
f = torch.exp(0.05 * m - 0.3)
eng = gp.get_engine(N, d)
m_new = torch.empty(N, dtype=torch.float64, device=dev); V_new = torch.empty((N, N), dtype=torch.float64, device=dev)
lib = _lib.load()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    rc = lib.gpfit_estep(eng._ctx, gp._stream(), K.data_ptr(), K.stride(0), N, r.data_ptr(), m.data_ptr(), f.data_ptr(),
                         syn.F_PARAMS["logA"], m_new.data_ptr(), V_new.data_ptr(), V_new.stride(0))
    torch.cuda.synchronize(); print(f"gpfit_estep N={N}: rc={rc} {1e3*(time.time()-t0):.2f} ms  ({2.667*N**3/(time.time()-t0)/1e12:.1f} TFLOP/s on 2.67 N^3)", flush=True)

# (4) one active-learning scoring step: 3000 candidates against N=4096 inducing points
N, d, ns = 4096, 256, 3000
grid = syn.grid_for(d)
th = tth(syn.theta_eval())
X = torch.from_numpy(syn.stimuli(N, d)).to(dev)
Xs = torch.from_numpy(np.random.default_rng(3).standard_normal((ns, d))).to(dev)
C, mask = gp.localker(th, upper, lower, grid)
Kt = gp.acosker(th, X, X, C=C)
r_np, m_np = syn.cell_inputs(N)
m = torch.from_numpy(m_np).to(dev); V = 0.5 * Kt
A, lam0 = float(np.exp(syn.F_PARAMS["logA"])), syn.F_PARAMS["lambda0"]
rr = torch.arange(0, 100, dtype=torch.float64, device=dev)
Kt_inv = gp.spd_inverse(Kt)          # model state in the notebook (K_tilde_inv_b), not part of the scoring step
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    Kvec_s = gp.acosker(th, Xs, x2=None, C=C, dC=None, diag=True)
    K_s = gp.acosker(th, Xs, x2=X, C=C, dC=None, diag=False)
    mu, s2 = gp.lambda_moments(Xs, Kt, gp.matmul(K_s, Kt_inv), Kvec_s, K_s, C, m, V, th)
    torch.cuda.synchronize(); t1 = time.time()
    u = gp.nd_utility(A ** 2 * s2, A * mu + lam0, rr)
    best = int(u.argmax()); torch.cuda.synchronize(); t2 = time.time()
    print(f"active-learning scoring, {ns} candidates x N={N}: kernel rows + lambda moments {1e3*(t1-t0):.2f} ms, utility+argmax {1e3*(t2-t1):.3f} ms, best {best}", flush=True)

# (5) sparse regime: nt = 4096 training points, n_tilde = 2048 inducing points, d = 256
nt_, ntil, d = 4096, 2048, 256
grid = syn.grid_for(d)
X = torch.from_numpy(syn.stimuli(nt_, d)).to(dev)
xt = X[:ntil].contiguous()
r_np, _ = syn.cell_inputs(nt_)
r = torch.from_numpy(r_np).to(dev)
th0 = tth(syn.theta0())
C, mask = gp.localker(th0, upper, lower, grid)
Kt0 = gp.acosker(th0, xt, xt, C=C)
gp.EIGVAL_TOL = 1e-4
ev, evec, keep = gp._eigen_stabilise(Kt0)
B = evec[:, keep].contiguous()
m_b = 0.1 * torch.randn(B.shape[1], dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
V_b = torch.diag(ev[keep]) * 0.5
fp = {"logA": torch.tensor(syn.F_PARAMS["logA"]), "lambda0": torch.tensor(syn.F_PARAMS["lambda0"])}
th1 = tth(syn.theta_eval())
print(f"sparse regime nt={nt_} n_tilde={ntil}: kept {B.shape[1]}", flush=True)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        loss, grad = gp._closure_general(th1, (lower, upper), grid, X, xt, r, B, m_b, V_b, fp, ntil, nt_)
        torch.cuda.synchronize(); print(f"  general closure (sparse): {time.time()-t0:.4f} s  loss {loss:.6f}", flush=True)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        loss2, grad2 = gp._closure_sparse(th1, (lower, upper), grid, X, xt, r, B, m_b, V_b, fp)
        torch.cuda.synchronize(); print(f"  sparse adjoint closure: {time.time()-t0:.4f} s  loss {loss2:.6f}  max grad dev "
                                        f"{max(abs(grad[k]-grad2[k]) for k in grad)/max(abs(v) for v in grad.values()):.2e}", flush=True)
