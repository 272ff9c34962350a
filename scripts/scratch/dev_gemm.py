"""Dev check on the GPU box: dgemm correctness (all layouts / tri flags / edges) + throughput."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaussian_processes_amd.build import lib_path
lib = ctypes.CDLL(lib_path())
lib.gpfit_last_error.restype = ctypes.c_char_p
vp, i32, i64, f64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double
lib.gpfit_dgemm.argtypes = [vp, i32, i32, i32, i32, i32, f64, vp, i64, vp, i64, f64, vp, i64, i32, i32, i32]
dev = torch.device("cuda:0")
torch.manual_seed(0)

def gemm(A, B, C, ak, bk, M, N, K, alpha=1.0, beta=0.0, lower=0, at=0, bt=0):
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.gpfit_dgemm(st, ak, bk, M, N, K, alpha, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), beta,
                         C.data_ptr(), C.stride(0), lower, at, bt)
    assert rc == 0, lib.gpfit_last_error()

def check(M, N, K, ak, bk, lower=0, at=0, bt=0, beta=0.0):
    opA = torch.randn(M, K, dtype=torch.float64, device=dev)
    opB = torch.randn(K, N, dtype=torch.float64, device=dev)
    if at == 1: opA = torch.tril(opA)
    if at == 2: opA = torch.triu(opA)
    if bt == 1: opB = torch.tril(opB)
    if bt == 2: opB = torch.triu(opB)
    A = opA.t().contiguous() if ak else opA.contiguous()
    B = opB.contiguous() if bk else opB.t().contiguous()
    C0 = torch.randn(M, N, dtype=torch.float64, device=dev)
    C = C0.clone()
    gemm(A, B, C, ak, bk, M, N, K, 0.7, beta, lower, at, bt)
    ref = 0.7 * (opA @ opB) + beta * C0
    if lower:
        T = 128
        mask = torch.zeros(M, N, dtype=torch.bool, device=dev)
        for ti in range((M + T - 1) // T):
            mask[ti*T:(ti+1)*T, :min(N, (ti+1)*T)] = True
        err = ((C - ref)[mask]).abs().max().item()
        untouched = (C[~mask] == C0[~mask]).all().item()
        assert untouched
    else:
        err = (C - ref).abs().max().item()
    tol = 1e-11 * max(1.0, K ** 0.5)
    print(f"M{M} N{N} K{K} ak{ak} bk{bk} lower{lower} at{at} bt{bt} beta{beta}: err {err:.2e}")
    assert err < tol, err

for ak in (0, 1):
    for bk in (0, 1):
        check(256, 384, 64, ak, bk)
        check(130, 70, 48, ak, bk, beta=0.5)
        check(64, 64, 16, ak, bk)
check(512, 512, 512, 0, 0, lower=1, beta=1.0)
check(512, 512, 512, 0, 1, at=1, bt=1)
check(512, 512, 512, 1, 1, at=2, bt=1)
check(384, 384, 384, 1, 1, lower=1, at=2, bt=1)
check(384, 384, 384, 0, 0, lower=1, bt=2)
print("correctness OK")

# throughput
for (M, N, K, ak, bk) in [(8192, 8192, 8192, 0, 0), (2048, 2048, 2048, 0, 0), (1024, 1024, 1024, 0, 0), (512, 512, 512, 0, 0), (256, 256, 256, 0, 1), (128, 128, 128, 0, 0)]:
    A = torch.randn(M if not ak else K, K if not ak else M, dtype=torch.float64, device=dev)
    B = torch.randn(K if bk else N, N if bk else K, dtype=torch.float64, device=dev)
    C = torch.empty(M, N, dtype=torch.float64, device=dev)
    for _ in range(2): gemm(A, B, C, ak, bk, M, N, K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps): gemm(A, B, C, ak, bk, M, N, K)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"dgemm M{M} N{N} K{K} ak{ak} bk{bk}: {ms:.3f} ms  {2.0*M*N*K/ms/1e9:.1f} TFLOP/s")
    if M == 8192 and K == 8192 and ak == 0 and bk == 0:
        A2 = A.clone()
        for _ in range(2): torch.matmul(A, B.t(), out=C)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): torch.matmul(A, B.t(), out=C)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"  rocBLAS (torch.matmul) same shape: {ms:.3f} ms  {2.0*M*N*K/ms/1e9:.1f} TFLOP/s")

# probes
lib.gpfit_probe_mfma_f64.argtypes = [vp, vp, i32, i32]
lib.gpfit_probe_stream_copy.argtypes = [vp, vp, vp, i64]
scr = torch.empty(2048 * 256, dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for blocks in (256, 512, 1024, 2048):
    iters = 20000
    lib.gpfit_probe_mfma_f64(st, scr.data_ptr(), blocks, 100); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lib.gpfit_probe_mfma_f64(st, scr.data_ptr(), blocks, iters); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    fl = blocks * 4 * iters * 8 * 2048.0
    print(f"mfma f64 probe blocks {blocks}: {ms:.2f} ms  {fl/ms/1e9:.1f} TFLOP/s")
n = 1 << 28
a = torch.randn(n, dtype=torch.float64, device=dev); b = torch.empty_like(a)
lib.gpfit_probe_stream_copy(st, a.data_ptr(), b.data_ptr(), n); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): lib.gpfit_probe_stream_copy(st, a.data_ptr(), b.data_ptr(), n)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"stream copy 2 GiB+2 GiB: {ms:.3f} ms  {2*n*8/ms/1e9:.2f} TB/s")
