"""The last stretch of a rocprofv3 --kernel-trace CSV: where the time of a grouped run goes.

    python scripts/trace_tail.py <kernel_trace.csv> <window_ms> [list_from_ms list_to_ms]

Per kernel class over the window that ends with the last kernel: calls, total, average, and the idle time of the queue
(gaps between consecutive kernels); optionally the launches between two offsets of the window, one per line."""
import collections
import csv
import sys

path, win = sys.argv[1], float(sys.argv[2])
t = list(csv.DictReader(open(path)))
for r in t:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
t.sort(key=lambda r: r['s'])
end = max(r['e'] for r in t)
t0 = end - int(win * 1e6)
w = [r for r in t if r['s'] >= t0]


def short(r):
    n = r['Kernel_Name'].split('(')[0].replace('void gpfit::', '')
    return n.replace('double', 'd').replace('float', 'f').replace('false', '0').replace('true', '1').replace(' ', '')[:52]


agg = collections.defaultdict(lambda: [0, 0.0])
busy, gaps, last = 0.0, 0.0, None
for r in w:
    d = (r['e'] - r['s']) / 1e3
    agg[short(r)][0] += 1
    agg[short(r)][1] += d
    busy += d
    if last is not None and r['s'] > last:
        gaps += (r['s'] - last) / 1e3
    last = max(last or 0, r['e'])
print(f"window {win} ms: {len(w)} kernels, kernel time {busy / 1e3:.2f} ms, idle between kernels {gaps / 1e3:.2f} ms")
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:54s} n {c:5d} total {d / 1e3:8.3f} ms avg {d / c:8.1f} us")
# the idle stretches: after which kernel class the queue waits, and for how long in total
idle = collections.defaultdict(lambda: [0, 0.0])
prev = None
for r in w:
    if prev is not None and r['s'] > prev['e'] + 1500:
        k = short(prev) + "  ->  " + short(r)
        idle[k][0] += 1
        idle[k][1] += (r['s'] - prev['e']) / 1e3
    if prev is None or r['e'] > prev['e']:
        prev = r
print("idle stretches > 1.5 us, by the kernels on either side:")
for k, (c, d) in sorted(idle.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {k:100s} n {c:4d} total {d / 1e3:7.3f} ms avg {d / c:7.1f} us")
if len(sys.argv) > 4:
    a, b = float(sys.argv[3]), float(sys.argv[4])
    prev = None
    for r in w:
        off = (r['s'] - t0) / 1e6
        if a <= off <= b:
            gap = (r['s'] - prev) / 1e3 if prev else 0.0
            print(f"{off:9.3f} ms {(r['e'] - r['s']) / 1e3:8.1f} us gap {gap:6.1f}  {short(r)}")
        prev = r['e']
