#!/bin/bash
# Headline unit under the schedule / hardware-queue settings that matter (bench.py itself, so the streams are created
# in the order the driver's run creates them): bash scripts/scratch/dev_headline_matrix.sh > gpurun_out/<tag>.log
cd "$GRAFT_REPO_ROOT" || exit 1
for q in 1 2 4 8; do
  for ls in 0 1; do
    v=$(GPU_MAX_HW_QUEUES=$q GPFIT_LOCKSTEP=$ls timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-in-flight 2>/dev/null | python -c "import json,sys; o=json.loads(sys.stdin.read()); r=o['roofline']; print(o['ms_per_step'], r['phases_ms']['chol_K_done'], r['phases_ms']['W_done'], r['achieved'])")
    echo "queues $q lockstep $ls: ms_per_step chol_K_done W_done T_tflops = $v"
  done
done
