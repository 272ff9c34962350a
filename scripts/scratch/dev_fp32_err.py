"""Where does the fp32 instance's loss error come from?  (a) rounding K~ and V to fp32 only, all
arithmetic fp64; (b) the fp32 instance.  Corner 448 of the theta lattice at N=8192."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaussian_processes_amd import synthetic as syn, utils as gp
from gaussian_processes_amd.engine import GPFitEngine
dev = torch.device("cuda:0")
N, d = 8192, 256
grid = syn.grid_for(d); lower, upper = syn.limits()
X = torch.from_numpy(syn.stimuli(N, d)).to(dev)
r_np, m_np = syn.cell_inputs(N); r, m = torch.from_numpy(r_np).to(dev), torch.from_numpy(m_np).to(dev)
t0 = {k: torch.tensor(v, dtype=torch.float64) for k, v in syn.theta0().items()}
C, mask = gp.localker(t0, upper, lower, grid); V = 0.5 * gp.acosker(t0, X, X, C=C)
pts = syn.theta_grid(8)
eng = GPFitEngine(N, d)
for pi in (448, 0, 511, 219):
    th = pts[pi]
    a = eng.fit_eval(th, lower, upper, grid, X, r, m, V, syn.F_PARAMS["logA"], syn.F_PARAMS["lambda0"], want_vectors=False)
    b = eng.fit_eval(th, lower, upper, grid, X.float(), r.float(), m.float(), V.float(), syn.F_PARAMS["logA"], syn.F_PARAMS["lambda0"], want_vectors=False)
    tt = {k: torch.tensor(v, dtype=torch.float64) for k, v in th.items()}
    Ct, _ = gp.localker(tt, upper, lower, grid); K = gp.acosker(tt, X, X, C=Ct)
    def parts(Kx, Vx, mx):
        L = torch.linalg.cholesky(Kx); LV = torch.linalg.cholesky(Vx)
        ldK = 2 * torch.log(torch.diagonal(L)).sum(); ldV = 2 * torch.log(torch.diagonal(LV)).sum()
        y = torch.linalg.solve_triangular(L, mx[:, None], upper=False)[:, 0]
        T = torch.linalg.solve_triangular(L, LV, upper=False)
        return float(ldK), float(ldV), float(y @ y), float((T * T).sum())
    p64 = parts(K, V, m)
    p32 = parts(K.float().double(), V.float().double(), m.float().double())
    # rounding X only (what an fp32 kernel build with exact arithmetic would see)
    Xr = X.float().double(); Kxr = gp.acosker(tt, Xr, Xr, C=Ct)
    pxr = parts(Kxr, V, m)
    KL64 = -0.5 * p64[1] + 0.5 * p64[0] + 0.5 * p64[2] + 0.5 * p64[3]
    print(f"point {pi}: loss fp64 {a['loss']:.6f} fp32 {b['loss']:.6f} rel {abs(a['loss']-b['loss'])/abs(a['loss']):.2e}; parts fp64 (ldK, ldV, mKm, tr) {p64}")
    print(f"   fp32 instance parts: ldK {b['logdet_K']:.4f} ldV {b['logdet_V']:.4f} mKm {b['mKinvm']:.5f} tr {b['tr_KinvV']:.4f} loglik {b['loglik']:.5f} (fp64 {a['loglik']:.5f})")
    print(f"   K~,V,m rounded to fp32, fp64 arithmetic: d ldK {p32[0]-p64[0]:+.4f} d ldV {p32[1]-p64[1]:+.4f} d mKm {p32[2]-p64[2]:+.5f} d tr {p32[3]-p64[3]:+.4f}")
    print(f"   X rounded to fp32, K~ built in fp64:     d ldK {pxr[0]-p64[0]:+.4f} d mKm {pxr[2]-p64[2]:+.5f} d tr {pxr[3]-p64[3]:+.4f}")
    print(f"   fp32 instance minus fp64:                d ldK {b['logdet_K']-a['logdet_K']:+.4f} d ldV {b['logdet_V']-a['logdet_V']:+.4f} d mKm {b['mKinvm']-a['mKinvm']:+.5f} d tr {b['tr_KinvV']-a['tr_KinvV']:+.4f} d loglik {b['loglik']-a['loglik']:+.5f}")
