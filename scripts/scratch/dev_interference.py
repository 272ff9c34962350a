"""How much does a concurrent long GEMM on another stream slow the latency-bound Cholesky chain?"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaussian_processes_amd import _lib, utils as gp, synthetic as syn
lib = _lib.load(); dev = torch.device("cuda:0")
n = 8192
g = torch.Generator().manual_seed(0)
M = torch.randn(n, n, dtype=torch.float64, generator=g).to(dev)
S = M @ M.T + n * torch.eye(n, dtype=torch.float64, device=dev)
A = torch.randn(n, n, dtype=torch.float64, device=dev); B = torch.randn(n, n, dtype=torch.float64, device=dev); C = torch.empty_like(A)
side = torch.cuda.Stream()
def potrf_ms(reps=3):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        L, Li, ld, info = gp.cholesky(S, want_inverse=True)
        ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts)
def background(walk, count):
    for _ in range(count):
        lib.gpfit_dgemm_ex(ctypes.c_void_p(side.cuda_stream), 0, 1, n, n, n, 1.0, A.data_ptr(), n, B.data_ptr(), n, 0.0, C.data_ptr(), n, 0, 0, 0, walk, 0)
print(f"potrf + inverse alone: {potrf_ms():.2f} ms")
for name, walk in (("full occupancy", 0), ("one workgroup per CU", 16)):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record(side); background(walk, 1); e1.record(side)
    torch.cuda.synchronize(); solo = e0.elapsed_time(e1)
    with torch.cuda.stream(side):
        e0.record(side); background(walk, 6); e1.record(side)
    t = potrf_ms(1)
    torch.cuda.synchronize()
    print(f"background dense 8192^3 GEMMs at {name} ({solo:.1f} ms each alone): potrf {t:.2f} ms; 6 GEMMs took {e0.elapsed_time(e1):.1f} ms")
