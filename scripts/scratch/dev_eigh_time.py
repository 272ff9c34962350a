import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaussian_processes_amd import utils as gp, synthetic as syn
dev = torch.device("cuda")
lower, upper = syn.limits()
th = {k: torch.tensor(v, dtype=torch.float64) for k, v in syn.theta0().items()}
for N in (2048, 4096, 8192):
    X = torch.from_numpy(syn.stimuli(N, 256)).to(dev)
    C, mask = gp.localker(th, upper, lower, 16)
    K = gp.acosker(th, X, X, C=C)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        ev, evec = torch.linalg.eigh(K)
        torch.cuda.synchronize(); t1 = time.time()
        L, Li, logdet, info = gp.cholesky(K, want_inverse=True)
        torch.cuda.synchronize(); t2 = time.time()
    print(f"N={N}: torch.linalg.eigh {t1-t0:.3f} s; gp.cholesky(+inverse) {t2-t1:.4f} s; lambda min {float(ev[0]):.3e} max {float(ev[-1]):.3e}; "
          f"1/tr(K^-1) = {1.0/float((Li*Li).sum()):.3e}", flush=True)
