run() { echo "== $*"; env "$@" python scripts/scratch/dev_phases.py 8192 256 6 2>&1 | grep -v amdgpu.ids | head -3 | cut -c1-200; }
run A=1
run GPFIT_T128_MIN=256 GPFIT_SK_ALL=1 GPFIT_SK_MIN_TILES=128
run GPFIT_T128_MIN=128 GPFIT_SK_ALL=1 GPFIT_SK_MIN_TILES=128
run GPFIT_T128_MIN=256 GPFIT_SK_ALL=1 GPFIT_SK_MIN_TILES=256
run GPFIT_T64_MIN=128
run GPFIT_DEEP_MAX=1024
run GPFIT_DEEP_MAX=256
