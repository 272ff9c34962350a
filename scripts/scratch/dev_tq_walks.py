"""T = L^-1 L_V and Q = I - T T^T shapes: stream-K tile-walk variants (dev)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaussian_processes_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
n = 8192
A = torch.tril(torch.randn(n, n, dtype=torch.float64, device=dev)); B = torch.tril(torch.randn(n, n, dtype=torch.float64, device=dev))
C = torch.empty(n, n, dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run(name, ak, bk, lower, at, bt, walk):
    best = 1e9
    for it in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.gpfit_dgemm_ex(st, ak, bk, n, n, n, 1.0, A.data_ptr(), n, B.data_ptr(), n, 0.0, C.data_ptr(), n, lower, at, bt, walk, 0)
        e1.record(); torch.cuda.synchronize(); assert rc == 0
        best = min(best, e0.elapsed_time(e1))
    print(f"{name:40s} walk {walk}: {best:7.3f} ms {n**3/3/best/1e9:6.1f} TF/s", flush=True)
for walk in (1, 5, 4, 2, 6, 3, 7):
    run("T  (0,1) lower a_tri1 b_tri1", 0, 1, 1, 1, 1, walk)
    run("Q  (0,0) lower a_tri1 b_tri2", 0, 0, 1, 1, 2, walk)
    run("W  (1,1) lower a_tri2 (dense B)", 1, 1, 1, 2, 0, walk)
