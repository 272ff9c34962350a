run() { echo "== $*"; env "$@" python bench.py --config n4096 --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('n4096', d['value'], d['ms_per_step'])"; env "$@" python bench.py --config cells64 --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cells64', d['value'], d['ms_per_step'])"; }
run A=1
env python bench.py --steps 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
for n in 2048 6144 12288; do python bench.py --n $n --steps 6 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=$n', d['value'], d['ms_per_step'])"; done
