for cfg in "0 0" "0 1" "2048 2" "2048 3" "1024 3" "4096 3" "512 3"; do
  set -- $cfg
  echo "== side_min=$1 half_occ=$2"; GPFIT_SIDE_MIN=$1 GPFIT_HALF_OCC=$2 python scripts/scratch/dev_phases.py 8192 256 6 2>&1 | grep -v amdgpu.ids
done
