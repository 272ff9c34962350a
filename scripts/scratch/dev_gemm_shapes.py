"""Product GEMM entry point on the shapes of the fit's large launches, standalone (compare with
scripts/scratch/dev_gemm_abl.hip and with the in-fit durations of the kernel trace)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaussian_processes_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
n = 8192
A = torch.randn(n, n, dtype=torch.float64, device=dev); B = torch.randn(n, n, dtype=torch.float64, device=dev)
C = torch.empty(n, n, dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run(name, ak, bk, M, N, K, lower, at, bt, walk, flops, rand=True):
    best = 1e9
    for it in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.gpfit_dgemm_ex(st, ak, bk, M, N, K, 1.0, A.data_ptr(), n, B.data_ptr(), n, 0.0, C.data_ptr(), n, lower, at, bt, walk, 0)
        e1.record(); torch.cuda.synchronize()
        assert rc == 0
        best = min(best, e0.elapsed_time(e1))
    print(f"{name:46s} {best:8.3f} ms {flops/best/1e9:6.1f} TF/s", flush=True)
h = 4096
run("8192 plain ak0 bk1", 0, 1, n, n, n, 0, 0, 0, 0, 2.0*n**3)
run("8192 R-type (1,1) b_tri=1 walk2", 1, 1, n, n, n, 0, 0, 1, 2, 1.0*n**3)
run("4096 plain (1,1) [Z21]", 1, 1, h, h, h, 0, 0, 0, 0, 2.0*h**3)
run("4096 (0,1) b_tri=1 walk2 [H]", 0, 1, h, h, h, 0, 0, 1, 2, 1.0*h**3)
run("4096 (1,1) a_tri=2 walk0 [W21]", 1, 1, h, h, h, 0, 2, 0, 0, 1.0*h**3)
run("4096 (1,1) lower uniform [syr2k]", 1, 1, h, h, h, 1, 0, 0, 0, 1.0*h**3)
run("4096 (0,0) b_tri=2 walk3 [trsm]", 0, 0, h, h, h, 0, 0, 2, 3, 1.0*h**3)
run("4096 (0,0) lower uniform [syrk]", 0, 0, h, h, h, 1, 0, 0, 0, 1.0*h**3)
run("8192 (0,1) both-tri lower rev [T]", 0, 1, n, n, n, 1, 1, 1, 1, n**3/3.0)
run("8192 (0,0) lower a_tri1 b_tri2 rev [Q]", 0, 0, n, n, n, 1, 1, 2, 1, n**3/3.0)
# zeros: power / data dependence
A.zero_(); B.zero_()
run("8192 plain ak0 bk1, ALL-ZERO operands", 0, 1, n, n, n, 0, 0, 0, 0, 2.0*n**3)
A.fill_(1.0); B.fill_(0.5)
run("8192 plain ak0 bk1, constant operands", 0, 1, n, n, n, 0, 0, 0, 0, 2.0*n**3)
