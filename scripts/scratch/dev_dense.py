"""One dense 8192^3 fp64 GEMM (NT) + syrk + R-shape: target of PMC passes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaussian_processes_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
N = 8192
A = torch.randn(N, N, dtype=torch.float64, device=dev)
B = torch.randn(N, N, dtype=torch.float64, device=dev)
C = torch.zeros(N, N, dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for rep in range(2):
    lib.gpfit_dgemm_ex(st, 0, 0, N, N, N, 1.0, A.data_ptr(), N, B.data_ptr(), N, 0.0, C.data_ptr(), N, 0, 0, 0, 0, 128)
    lib.gpfit_dgemm_ex(st, 0, 0, N, N, N, 1.0, A.data_ptr(), N, B.data_ptr(), N, 0.0, C.data_ptr(), N, 1, 0, 0, 0, 128)
    lib.gpfit_dgemm_ex(st, 1, 1, N, N, N, 1.0, A.data_ptr(), N, B.data_ptr(), N, 0.0, C.data_ptr(), N, 0, 0, 1, 2, 128)
torch.cuda.synchronize()
