import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaussian_processes_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
scr = torch.empty(4096 * 256, dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for blocks in (256, 512, 1024, 2048):
    iters = 10000
    lib.gpfit_probe_mfma_f64(st, scr.data_ptr(), blocks, 100); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lib.gpfit_probe_mfma_f64(st, scr.data_ptr(), blocks, iters); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1); fl = blocks * 4 * iters * 16 * 2048.0
    print(f"mfma f64 probe blocks {blocks} ({blocks/256:.0f} waves/SIMD): {ms:.2f} ms  {fl/ms/1e9:.1f} TFLOP/s")
