"""In-kernel shader clock of the dominant GEMM launches in the steady-state fit loop (diagnostic build,
scripts/scratch/dev_gemm_clock.sh): >= 2 s of back-to-back fits on the bench inputs, then the s_memtime /
s_memrealtime stamps of the last T (B k-major) and Q launches -- median over workgroups -- beside the
HIP-event duration of those launches.  MI355X_MICROARCH.md 'DVFS give-back' item 6."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gaussian_processes_amd import _lib
_lib.lib_path = lambda: os.path.join(ROOT, "gpurun_tmp", "libgpfit_clk.so")
from gaussian_processes_amd import synthetic as syn
from gaussian_processes_amd.engine import GPFitEngine
import bench

N, d = 8192, 256
dev = torch.device("cuda:0")
grid = syn.grid_for(d)
X = torch.from_numpy(syn.stimuli(N, d)).to(dev)
r_np, m_np = syn.cell_inputs(N)
r, m = torch.from_numpy(r_np).to(dev), torch.from_numpy(m_np).to(dev)
V = bench.build_V(X, grid, syn.theta0(), dev)
lower, upper = syn.limits()
eng = GPFitEngine(N, d)
th = syn.theta_eval()
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    eng.fit_eval(th, lower, upper, grid, X, r, m, V, syn.F_PARAMS["logA"], syn.F_PARAMS["lambda0"], want_vectors=False); n += 1
torch.cuda.synchronize()
print(f"{n} fits in {time.time()-t0:.2f} s -> {(time.time()-t0)/n*1e3:.2f} ms/fit")
lib = _lib.load()
lib.gpfit_dev_gemm_clock.restype = ctypes.c_int
buf = (ctypes.c_longlong * (2 * 4096 * 2))()
assert lib.gpfit_dev_gemm_clock(buf) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(2, 4096, 2)
for which, name in ((1, "T = L^-1 L_V   (gemm_epi_kernel<double,false,true,2>)"), (0, "Q = I - T T^T   (gemm_epi_kernel<double,false,false,1>)")):
    st = a[which]; ok = st[:, 1] > 0
    ghz = st[ok, 0] / (st[ok, 1] * 10.0)
    print(f"{name}: {ok.sum()} workgroups, in-kernel clock median {np.median(ghz):.3f} GHz (p10 {np.percentile(ghz,10):.3f}, p90 {np.percentile(ghz,90):.3f}); "
          f"median workgroup {np.median(st[ok,1])*10/1e3:.1f} us, {np.median(st[ok,0]):.0f} cycles")
eng.close()
