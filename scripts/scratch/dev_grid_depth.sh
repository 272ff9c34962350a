#!/bin/bash
# theta-grid throughput against the number of points kept in flight (mixed and fp64, 96 points)
cd "$GRAFT_REPO_ROOT" || exit 1
for dt in mixed f64 f32; do for dep in 1 2 3 4; do
  python bench.py --config thetagrid --dtype $dt --grid-points 96 --depth $dep --steps 1 --warmup 0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$dt depth $dep:', j['value'], j['unit'])"
done; done
