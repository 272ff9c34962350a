#!/bin/bash
# One-line-per-setting timing of the headline unit under tuning knobs given as arguments, e.g.
#   bash scripts/scratch/dev_knobs.sh "" "GPFIT_FORK_EARLY=1" "GPFIT_SIDE_MIN=2048"
cd "$GRAFT_REPO_ROOT" || exit 1
for kv in "$@"; do
  echo "[$kv]"
  env $kv python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-in-flight 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['phases_ms'])"
done
