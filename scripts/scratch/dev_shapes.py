"""Micro-benchmark of the big triangular GEMM shapes of one fit (N=8192) with walk/tile variants."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaussian_processes_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
A = torch.randn(N, N, dtype=torch.float64, device=dev)
B = torch.randn(N, N, dtype=torch.float64, device=dev)
C = torch.zeros(N, N, dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream

def run(name, ak, bk, lower, at, bt, flops, walks=(0, 1, 2, 3), tiles=(128, 64)):
    for tile in tiles:
        for w in walks:
            if lower and (w & 2): continue
            f = lambda: lib.gpfit_dgemm_ex(st, ak, bk, N, N, N, 1.0, A.data_ptr(), N, B.data_ptr(), N, 0.0, C.data_ptr(), N, lower, at, bt, w, tile)
            f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): f()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            print(f"{name:28s} tile {tile:3d} walk {w}: {ms:7.3f} ms  {flops/ms/1e9:6.1f} TFLOP/s", flush=True)

n3 = float(N) ** 3
run("dense NT (syrk-like full)", 0, 0, 0, 0, 0, 2 * n3, walks=(0, 4), tiles=(128,))
run("syrk lower NT", 0, 0, 1, 0, 0, n3, walks=(0, 4), tiles=(128,))
run("T = Li*LV  NN lower tri tri", 0, 1, 1, 1, 1, n3 / 3)
run("P = T*T^T  NT lower tri tri", 0, 0, 1, 1, 2, n3 / 3)
run("R = Q*Li   TN dense b_tri=1", 1, 1, 0, 0, 1, n3, walks=(2, 6), tiles=(128,))
run("W = Li^T*R TN lower a_tri=2", 1, 1, 1, 2, 0, n3 / 3)
run("trsm L21=A21*Li^T NT b_tri=2", 0, 0, 0, 0, 2, n3)
run("Li21 = Li22*tmp NN a_tri=1", 0, 1, 0, 1, 0, n3)
