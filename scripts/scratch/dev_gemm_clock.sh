#!/bin/bash
# Diagnostic build of the library with in-kernel clock stamps around the scheduled GEMM body
# (gemm.hip, GPFIT_CLOCK_STAMPS), into gpurun_tmp/libgpfit_clk.so.  Run here; then on the box:
#   python scripts/scratch/dev_gemm_clock.py
set -e
cd "$(dirname "$0")/.."
python -m gaussian_processes_amd.build > /dev/null
mkdir -p gpurun_tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Iinclude -Igaussian_processes_amd/csrc \
  -DGPFIT_CLOCK_STAMPS -c gaussian_processes_amd/csrc/gemm.hip -o gpurun_tmp/gemm_clk.o
objs=$(ls gaussian_processes_amd/lib/*.o | grep -v "/gemm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_tmp/libgpfit_clk.so $objs gpurun_tmp/gemm_clk.o
ls -la gpurun_tmp/libgpfit_clk.so
