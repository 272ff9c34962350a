"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals and, for one steady-state fit,
per-queue busy time, overlap and the largest launches."""
import collections, csv, sys

path = sys.argv[1]
fit_index = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t = list(csv.DictReader(open(path)))
for r in t:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
t.sort(key=lambda r: r['s'])
starts = [i for i, r in enumerate(t) if 'pack_lower' in r['Kernel_Name']]
fit = t[starts[fit_index]:starts[fit_index + 1]]
T0 = min(r['s'] for r in fit); T1 = max(r['e'] for r in fit)
print(f"fit wall {(T1-T0)/1e6:.3f} ms, {len(fit)} kernels")
byq = collections.defaultdict(list)
for r in fit: byq[r['Queue_Id']].append(r)
for q, rs in byq.items():
    busy = sum(r['e'] - r['s'] for r in rs)
    print(f"queue {q}: n {len(rs)} busy {busy/1e6:.3f} ms span {(max(r['e'] for r in rs)-min(r['s'] for r in rs))/1e6:.3f} ms first {(min(r['s'] for r in rs)-T0)/1e6:.3f}")
ev = sorted([(r['s'], 1) for r in fit] + [(r['e'], -1) for r in fit])
cur = 0; last = None; tot = collections.Counter()
for ts, dv in ev:
    if last is not None: tot[min(cur, 2)] += ts - last
    cur += dv; last = ts
print(f"idle {tot[0]/1e6:.3f} ms, one {tot[1]/1e6:.3f} ms, >=2 {tot[2]/1e6:.3f} ms")
agg = collections.defaultdict(lambda: [0, 0])
for r in fit:
    name = r['Kernel_Name'].split('(')[0][-48:]
    agg[name][0] += 1; agg[name][1] += r['e'] - r['s']
for name, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {name:50s} n {n:4d} total {d/1e6:8.3f} ms avg {d/n/1e3:8.1f} us")
# histogram of gemm durations by grid size
hist = collections.defaultdict(lambda: [0, 0])
for r in fit:
    if 'dgemm' in r['Kernel_Name']:
        g = int(r['Grid_Size_X']) // 256
        hist[g][0] += 1; hist[g][1] += r['e'] - r['s']
print("gemm by #blocks:")
for g, (n, d) in sorted(hist.items()):
    print(f"  blocks {g:6d} n {n:4d} total {d/1e6:7.3f} ms avg {d/n/1e3:8.1f} us")
print("launches > 0.3 ms, in start order:")
for r in fit:
    d = (r['e'] - r['s']) / 1e6
    if d > 0.3:
        print(f"  t={(r['s']-T0)/1e6:8.3f} dur {d:7.3f} q{r['Queue_Id']} blocks {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):6d} {r['Kernel_Name'].split('(')[0][-44:]}")
