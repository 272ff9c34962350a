#!/bin/bash
# each argument: "VAR=val VAR2=val2 ..." -> bench ms/step at the headline and at N=4096 d=128
for combo in "$@"; do
  a=$(env $combo python bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(env $combo python bench.py --no-cpu-baseline --steps 8 --warmup 2 --n 4096 --d 128 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "[$combo]  N8192: $a ms   N4096: $b ms"
done
