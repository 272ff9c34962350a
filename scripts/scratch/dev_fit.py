"""Dev check on the GPU box: gpfit_fit_eval vs the CPU oracle (Cholesky formulation)."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaussian_processes_amd import _lib, synthetic as syn
from oracle import gp_oracle as orc
lib = _lib.load()
dev = torch.device("cuda:0")
KEYS = syn.THETA_KEYS
lower, upper = syn.limits()

def run(N, d, reps=0, check=True):
    grid = syn.grid_for(d)
    X = torch.from_numpy(syn.stimuli(N, d))
    r_np, m_np = syn.cell_inputs(N)
    r, m = torch.from_numpy(r_np), torch.from_numpy(m_np)
    th0, th1 = syn.theta0(), syn.theta_eval()
    Xd, rd, md = X.to(dev), r.to(dev), m.to(dev)
    ctx = ctypes.c_void_p()
    _lib.check(lib.gpfit_ctx_create(0, N, d, d, ctypes.byref(ctx)), "ctx_create")
    # V = 0.5 K~(theta0) built by the oracle at small N, on the GPU by torch at large N
    if check:
        C0, mask0 = orc.spatial_metric(th0, lower, upper, grid)
        V = 0.5 * orc.arccos_gram(th0, X[:, mask0], X[:, mask0], C0)
        Vd = V.to(dev)
    else:
        C0, mask0 = orc.spatial_metric(th0, lower, upper, grid)
        Xg = Xd
        C0d = C0.to(dev)
        XC = Xg @ C0d
        q = torch.sqrt((XC * Xg).sum(1) + 1.0)
        G = XC @ Xg.T + 1.0
        c = torch.clip(G / (torch.outer(q, q) + 1e-7), -1, 1)
        dl = torch.arccos(c)
        Kt = torch.outer(q, q) * (torch.sqrt(1 - c * c) + orc.PI32 * c - dl * c) / orc.PI32
        Vd = 0.25 * (Kt + Kt.T)
        del G, c, dl, Kt, XC
    out = (ctypes.c_double * 16)()
    lam_m = torch.empty(N, dtype=torch.float64, device=dev); lam_var = torch.empty_like(lam_m); f = torch.empty_like(lam_m)
    th = _lib.darr([th1[k] for k in KEYS]); lo = _lib.darr([lower[k] for k in KEYS]); up = _lib.darr([upper[k] for k in KEYS])
    st = torch.cuda.current_stream().cuda_stream
    def call(grad=1):
        return lib.gpfit_fit_eval(ctx, st, th, lo, up, grid[0], grid[1], Xd.data_ptr(), Xd.stride(0), N, rd.data_ptr(), md.data_ptr(),
                                Vd.data_ptr(), Vd.stride(0), syn.F_PARAMS["logA"], syn.F_PARAMS["lambda0"], grad, out,
                                lam_m.data_ptr(), lam_var.data_ptr(), f.data_ptr())
    rc = call()
    print(f"N={N} d={d}: rc={rc} err='{_lib.last_error()}' out={[f'{v:.10g}' for v in out]}")
    if check:
        t0 = time.time()
        loss, grad, parts = orc.mstep_closure_cholesky(th1, lower, upper, grid, X, r, m, V, syn.F_PARAMS["logA"], syn.F_PARAMS["lambda0"], want_parts=True)
        print(f"  oracle {time.time()-t0:.2f}s loss {loss:.12g} loglik {parts['loglik']:.12g} KL {parts['KL']:.12g}")
        print("  rel err loss", abs(out[0] - loss) / abs(loss), "KL", abs(out[2] - parts['KL']) / abs(parts['KL']))
        g = np.array([grad[k] for k in KEYS]); gg = np.array(out[3:9])
        print("  grad oracle", g); print("  grad gpu   ", gg); print("  grad rel err", np.abs(g - gg).max() / np.abs(g).max())
        print("  lam_var rel", float((lam_var.cpu() - parts['lam_var']).abs().max() / parts['lam_var'].abs().max()),
              "f rel", float((f.cpu() - parts['f']).abs().max() / parts['f'].abs().max()))
    if reps:
        for g_ in (1, 0):
            call(g_); torch.cuda.synchronize(); t0 = time.time()
            for _ in range(reps): call(g_)
            torch.cuda.synchronize(); dt = (time.time() - t0) / reps
            F = (14 / 3) * N**3 + 4 * N * N * d + 4 * N * d * d
            print(f"  grad={g_}: {dt*1e3:.2f} ms/fit  {1/dt:.2f} fits/s  algorithmic {F/dt/1e12:.1f} TFLOP/s ({F/dt/78.6e12*100:.1f}% of 78.6)")
    lib.gpfit_ctx_destroy(ctx)

for N, d in [(64, 64), (200, 16), (256, 64), (512, 64), (1000, 100)]:
    run(N, d)
run(2048, 256, reps=3)
run(4096, 128, reps=3, check=False)
run(8192, 256, reps=3, check=False)
