for cfg in "4 4 8" "4 4 4" "4 4 16" "4 8 8" "8 4 8" "2 8 8" "8 2 8" "2 2 8" "4 2 8" "2 4 8" "8 8 4" "6 6 8" "3 3 8" "4 16 8" "16 4 8"; do
  set -- $cfg
  echo "== GR=$1 GC=$2 CUT=$3"
  GPFIT_XCD_GR=$1 GPFIT_XCD_GC=$2 GPFIT_XCD_CUT=$3 python scripts/scratch/dev_xcd.py 2>&1 | grep -v amdgpu.ids | awk '{ printf "%s ", $1; for (i=1;i<=NF;i++) if ($i=="TF/s") printf "%s ", $(i-1); print "" }'
done
