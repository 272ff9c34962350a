#!/bin/bash
# usage: dev_sweep_flag.sh VAR   -> bench ms/step with VAR unset and VAR=1
var=$1
for mode in unset set; do
  if [ $mode = set ]; then export $var=1; fi
  a=$(python bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], 'dom', j['roofline']['achieved'], 'fam', j['roofline']['gemm_family']['tflops'])")
  b=$(python bench.py --no-cpu-baseline --steps 8 --warmup 2 --n 4096 --d 128 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$var $mode  N8192: $a   N4096: $b ms"
done
