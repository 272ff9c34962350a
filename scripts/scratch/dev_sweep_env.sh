#!/bin/bash
# usage: dev_sweep_env.sh VAR v1 v2 ...   -> bench ms/step at the headline and at N=4096 d=128 for each value
var=$1; shift
for v in "$@"; do
  a=$(env $var=$v python bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(env $var=$v python bench.py --no-cpu-baseline --steps 8 --warmup 2 --n 4096 --d 128 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$var=$v  N8192: $a ms   N4096: $b ms"
done
