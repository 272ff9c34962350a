#!/bin/bash
# Sweep of the V-chain start offset (GPFIT_V_AFTER) on the headline unit of work.
cd "$GRAFT_REPO_ROOT" || exit 1
for v in 0 1024 2048 4096; do
  echo "GPFIT_V_AFTER=$v"
  GPFIT_V_AFTER=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['phases_ms'])"
done
