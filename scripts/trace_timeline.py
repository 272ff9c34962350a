"""Launch-by-launch timeline of one steady-state fit from a rocprofv3 --kernel-trace CSV.

    python scripts/trace_timeline.py <kernel_trace.csv> [fit_index] [min_us] [units_per_group]

Every launch of the fit in start order: start (ms from the fit's first kernel), duration, queue, workgroups,
short kernel name; then per queue and kernel class the totals.  Written for reading the factorisation
chains (which launches sit on the critical path and what the gaps between them are)."""
import collections
import csv
import sys

path = sys.argv[1]
fit_index = int(sys.argv[2]) if len(sys.argv) > 2 else 2
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
stride = int(sys.argv[4]) if len(sys.argv) > 4 else 1      # grouped evaluations: one "fit" = a group of this many units
t = list(csv.DictReader(open(path)))
for r in t:
    r['s'] = int(r['Start_Timestamp'])
    r['e'] = int(r['End_Timestamp'])
t.sort(key=lambda r: r['s'])
starts = [i for i, r in enumerate(t) if 'localker_kernel' in r['Kernel_Name']]
fit = t[starts[fit_index * stride]:starts[(fit_index + 1) * stride]]
T0 = min(r['s'] for r in fit)
T1 = max(r['e'] for r in fit)
qids = {}
for r in fit:
    qids.setdefault(r['Queue_Id'], len(qids))


def short(r):
    n = r['Kernel_Name'].split('(')[0].replace('void gpfit::', '')
    return n.replace('double', 'd').replace('float', 'f').replace('false', '0').replace('true', '1').replace(' ', '')[:48]


def blocks(r):
    gx = int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0)
    gy = int(r.get('Grid_Size_Y', 1) or 1)
    gz = int(r.get('Grid_Size_Z', 1) or 1)
    wx = max(1, int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1)) or 1))
    return gx // wx * max(1, gy) * max(1, gz)


print(f"fit wall {(T1 - T0) / 1e6:.3f} ms, {len(fit)} kernels, {len(qids)} queues")
last_end = {}
for r in fit:
    q = qids[r['Queue_Id']]
    d = (r['e'] - r['s']) / 1e3
    gap = (r['s'] - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = r['e']
    if d >= min_us:
        print(f"{(r['s'] - T0) / 1e6:8.3f} ms  {d:8.1f} us  gap {gap:7.1f}  q{q}  wg {blocks(r):5d}  {short(r)}")
agg = collections.defaultdict(lambda: [0, 0.0])
for r in fit:
    agg[(qids[r['Queue_Id']], short(r))][0] += 1
    agg[(qids[r['Queue_Id']], short(r))][1] += (r['e'] - r['s']) / 1e3
print("per queue and kernel:")
for (q, n), (c, d) in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[1][1])):
    print(f"  q{q} {n:50s} n {c:4d} total {d / 1e3:8.3f} ms avg {d / c:8.1f} us")
