"""XCD-aware schedule (walk bit 3) against the current walks on the fit's large launch shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaussian_processes_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
L1 = torch.tril(torch.randn(n, n, dtype=torch.float64, device=dev)); L2 = torch.tril(torch.randn(n, n, dtype=torch.float64, device=dev))
D = torch.randn(n, n, dtype=torch.float64, device=dev)
C = torch.empty(n, n, dtype=torch.float64, device=dev); C2 = torch.empty(n, n, dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
def call(A, B, Cout, ak, bk, M, N, K, lower, at, bt, walk):
    return lib.gpfit_dgemm_ex(st, ak, bk, M, N, K, 1.0, A.data_ptr(), n, B.data_ptr(), n, 0.0, Cout.data_ptr(), n, lower, at, bt, walk, 0)
def run(name, A, B, ak, bk, M, N, K, lower, at, bt, walks, flops, check=True):
    res = []
    for walk in walks:
        best = 1e9
        for it in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); rc = call(A, B, C if walk == walks[0] else C2, ak, bk, M, N, K, lower, at, bt, walk); e1.record(); torch.cuda.synchronize()
            assert rc == 0, _lib.last_error()
            best = min(best, e0.elapsed_time(e1))
        err = ""
        if check and walk != walks[0]:
            a, b = (torch.tril(C[:M, :N]), torch.tril(C2[:M, :N])) if lower else (C[:M, :N], C2[:M, :N])
            err = f" maxdiff {float((a - b).abs().max()):.2e}"
        res.append(f"walk {walk:2d}: {best:6.3f} ms {flops/best/1e9:5.1f} TF/s{err}")
    print(f"{name:44s} " + " | ".join(res), flush=True)
h = n // 2
f3 = n**3 / 3.0
run("T   (0,1) lower a_tri1 b_tri1 n", L1, L2, 0, 1, n, n, n, 1, 1, 1, (1, 8, 12), f3)
run("Q   (0,0) lower a_tri1 b_tri2 n", L1, L1, 0, 0, n, n, n, 1, 1, 2, (3, 8), f3)
run("W   (1,1) lower a_tri2 dense B n", L1, D, 1, 1, n, n, n, 1, 2, 0, (0, 8), f3)
run("R   (1,1) dense out b_tri1 n", D, L1, 1, 1, n, n, n, 0, 0, 1, (2, 8), 1.0 * n**3)
run("trsm (0,0) b_tri2 h", D, L1, 0, 0, h, h, h, 0, 0, 2, (3, 8), 1.0 * h**3)
run("syrk (0,0) lower dense h", D, D, 0, 0, h, h, h, 1, 0, 0, (0, 8), 1.0 * h**3)
run("tmp  (0,1) b_tri1 h", D, L1, 0, 1, h, h, h, 0, 0, 1, (2, 8), 1.0 * h**3)
run("Li21 (0,1) a_tri1 h", L1, D, 0, 1, h, h, h, 0, 1, 0, (1, 8), 1.0 * h**3)
run("Z21  (1,1) dense h", D, D, 1, 1, h, h, h, 0, 0, 0, (0, 8), 2.0 * h**3)
run("H    (0,1) b_tri1 h [walk2]", D, L1, 0, 1, h, h, h, 0, 0, 1, (2, 8), 1.0 * h**3)
run("W21  (1,1) a_tri2 h", L1, D, 1, 1, h, h, h, 0, 2, 0, (0, 8), 1.0 * h**3)
run("syr2k (1,1) lower dense h", D, D, 1, 1, h, h, h, 1, 0, 0, (0, 8), 1.0 * h**3)
run("dense n (0,1)", D, D, 0, 1, n, n, n, 0, 0, 0, (0, 8), 2.0 * n**3)
