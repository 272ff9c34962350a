"""GPU-busy share of a rocprofv3 --kernel-trace CSV over its last <window_s> seconds:
    python scripts/trace_busy.py <kernel_trace.csv> <window_s>
kernel time (union of the kernels' intervals), idle time, launches, and the ten kernel classes with the most time."""
import collections, csv, sys
path, win = sys.argv[1], float(sys.argv[2])
t = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(path))]
t.sort()
end = max(e for _, e, _ in t)
t0 = end - int(win * 1e9)
w = [x for x in t if x[0] >= t0]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in w:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = end - w[0][0]
print(f"window {span / 1e9:.3f} s: {len(w)} kernels, GPU busy {busy / 1e9:.3f} s = {busy / span:.2f} of the window")
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in w:
    k = n.split('(')[0].replace('void gpfit::', '').replace('void ', '')[:70]
    agg[k][0] += 1; agg[k][1] += e - s
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {k:72s} n {c:6d} total {d / 1e6:8.2f} ms avg {d / c / 1e3:7.1f} us")
