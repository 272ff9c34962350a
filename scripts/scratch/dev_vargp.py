import contextlib, io, os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaussian_processes_amd import utils as gp, synthetic as syn, _lib
g = np.load("tests/golden/g6_vargp_full_N128.npz")
KEYS = syn.THETA_KEYS; LOWER, UPPER = syn.limits()
T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64)).cuda()
th = {k: torch.tensor(float(v), requires_grad=True) for k, v in zip(KEYS, g["theta0"])}
X, r = T(g["X"]), T(g["r"])
gp.EIGVAL_TOL = 1e-14
C, mask = gp.localker(th, UPPER, LOWER, 8)
Kt = gp.acosker(th, X, X, C=C)
Kvec = gp.acosker(th, X, None, C=C, diag=True)
ev, evec = torch.linalg.eigh(Kt)
B = evec
m_b = torch.zeros(128, device="cuda"); V_b = torch.diag(ev)
fp = {"logA": torch.tensor(syn.F_PARAMS["logA"]), "lambda0": torch.tensor(syn.F_PARAMS["lambda0"])}
lm, lv = gp.lambda_moments(X, torch.diag(ev), B, Kvec, gp.matmul(Kt, B), C, m_b, V_b, th, kernfun=gp.acosker)
fp["lambda0"] = gp.lambda0_given_logA(fp["logA"], r, lm, lv)
f = gp.mean_f_given_lambda_moments(fp, lm, lv)
mg, Vg = gp.Estep(r=r, KKtilde_inv=B, m=m_b, f_params=fp, f_mean=f, K_tilde=torch.diag(ev), K_tilde_inv=torch.diag(1/ev))
eng = gp.get_engine(128, 1)
m_new = torch.empty(128, device="cuda"); V_new = torch.empty((128, 128), device="cuda")
m_orig = gp.matmul(B, m_b)
rc = _lib.load().gpfit_estep(eng._ctx, gp._stream(), Kt.data_ptr(), Kt.stride(0), 128, r.data_ptr(), m_orig.data_ptr(), f.data_ptr(), float(fp["logA"]), m_new.data_ptr(), V_new.data_ptr(), V_new.stride(0))
print("rc", rc)
mf = gp.matmul(B, m_new, transA=True); Vf = gp.matmul(B, gp.matmul(V_new, B), transA=True)
print("m diff", float((mf - mg).abs().max()), float(mg.abs().max()))
print("V diff", float((Vf - Vg).abs().max()), float(Vg.abs().max()))
# reference formula on CPU
Kc, fc, rc_, A = Kt.cpu(), f.cpu(), r.cpu(), float(torch.exp(fp["logA"]))
G = A * A * torch.diag(fc)
Vref = torch.linalg.solve(torch.eye(128) + Kc @ G, Kc)
mref = Vref @ (A * (rc_ - fc))
print("fused vs cpu V", float((V_new.cpu() - Vref).abs().max()), "m", float((m_new.cpu() - mref).abs().max()))
print("general vs cpu V", float(((B @ Vg @ B.T).cpu() - Vref).abs().max()))
print("---- second E-step with m != 0")
m_b, V_b = mg, Vg
f2, lm, lv = gp.mean_f(f_params=fp, calculate_moments=True, x=X, K_tilde=torch.diag(ev), KKtilde_inv=B, Kvec=Kvec, K=gp.matmul(Kt, B), C=C, m=m_b, V=V_b, theta=th, kernfun=gp.acosker)
mg2, Vg2 = gp.Estep(r=r, KKtilde_inv=B, m=m_b, f_params=fp, f_mean=f2, K_tilde=torch.diag(ev), K_tilde_inv=torch.diag(1/ev))
m_orig = gp.matmul(B, m_b)
rc = _lib.load().gpfit_estep(eng._ctx, gp._stream(), Kt.data_ptr(), Kt.stride(0), 128, r.data_ptr(), m_orig.data_ptr(), f2.data_ptr(), float(fp["logA"]), m_new.data_ptr(), V_new.data_ptr(), V_new.stride(0))
mf = gp.matmul(B, m_new, transA=True); Vf = gp.matmul(B, gp.matmul(V_new, B), transA=True)
print("m diff", float((mf - mg2).abs().max()), float(mg2.abs().max()))
print("V diff", float((Vf - Vg2).abs().max()), float(Vg2.abs().max()))
Gd = A * A * f2.cpu()
Vref = torch.linalg.solve(torch.eye(128) + Kc @ torch.diag(Gd), Kc)
mref = Vref @ (Gd * m_orig.cpu() + A * (rc_ - f2.cpu()))
print("fused vs cpu m", float((m_new.cpu() - mref).abs().max()), "general vs cpu m", float(((B @ mg2).cpu() - mref).abs().max()))
