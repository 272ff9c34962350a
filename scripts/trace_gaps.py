"""Per-queue timeline of one steady-state fit from a rocprofv3 --kernel-trace CSV: busy time, gaps
between consecutive kernels of a queue (launch-bound stretches show up as gaps well above the
~1.5 us dependent-launch boundary), and the phase boundaries (start of the factorisations, join)."""
import collections, csv, sys

path = sys.argv[1]
fit_index = int(sys.argv[2]) if len(sys.argv) > 2 else 2
t = list(csv.DictReader(open(path)))
for r in t:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
t.sort(key=lambda r: r['s'])
starts = [i for i, r in enumerate(t) if 'localker_kernel' in r['Kernel_Name']]
fit = t[starts[fit_index]:starts[fit_index + 1]]
T0 = min(r['s'] for r in fit); T1 = max(r['e'] for r in fit)
print(f"fit wall {(T1-T0)/1e6:.3f} ms, {len(fit)} kernels")
byq = collections.defaultdict(list)
for r in fit: byq[r['Queue_Id']].append(r)
for q, rs in byq.items():
    rs.sort(key=lambda r: r['s'])
    busy = sum(r['e'] - r['s'] for r in rs)
    gaps = [rs[i + 1]['s'] - rs[i]['e'] for i in range(len(rs) - 1)]
    big = [g for g in gaps if g > 3000]
    print(f"queue {q}: n {len(rs)} busy {busy/1e6:.3f} ms span {(rs[-1]['e']-rs[0]['s'])/1e6:.3f} ms first {(rs[0]['s']-T0)/1e6:.3f} "
          f"gaps total {sum(gaps)/1e6:.3f} ms median {sorted(gaps)[len(gaps)//2]/1e3:.2f} us; {len(big)} gaps > 3us totalling {sum(big)/1e6:.3f} ms")
    # gap histogram
    h = collections.Counter(min(int(g / 1000), 20) for g in gaps)
    print("   gap histogram (us: count):", " ".join(f"{k}:{h[k]}" for k in sorted(h)))
# timeline of phases on the main queue (the one holding the gram kernel)
mainq = next(q for q, rs in byq.items() if any('gram_acos' in r['Kernel_Name'] for r in rs))
rs = byq[mainq]
def short(r): return r['Kernel_Name'].split('(')[0].replace('void gpfit::', '')[:60]
marks = {}
for r in rs:
    n = r['Kernel_Name']
    if 'gram_acos' in n: marks['gram end'] = r['e']
    if 'chol_leaf' in n:
        marks.setdefault('first leaf', r['s']); marks['last leaf end'] = r['e']
    if 'frob_tile' in n: marks['frob (T done)'] = r['e']
    if 'adjoint_kernel' in n: marks['adjoint start'] = r['s']
for k, v in marks.items(): print(f"  {k:20s} t = {(v-T0)/1e6:8.3f} ms")
for q, rs in byq.items():
    if q == mainq: continue
    print(f"  aux queue {q}: {(rs[0]['s']-T0)/1e6:.3f} .. {(rs[-1]['e']-T0)/1e6:.3f} ms")
# within the K~ chain: time by kernel class between first leaf and last leaf
lo, hi = marks['first leaf'], marks['last leaf end']
cls = collections.defaultdict(lambda: [0, 0])
prev_e = None; gap_tot = 0
for r in rs if False else byq[mainq]:
    if r['s'] < lo or r['e'] > hi: continue
    n = r['Kernel_Name']
    key = 'leaf' if 'chol_leaf' in n else ('gemm128' if ', 128, ' in n or 'streamk' in n else ('gemm64' if ', 64, ' in n else ('gemm32' if ', 32, ' in n else 'other')))
    cls[key][0] += 1; cls[key][1] += r['e'] - r['s']
    if prev_e is not None: gap_tot += max(0, r['s'] - prev_e)
    prev_e = r['e']
print(f"K~ chain (first leaf .. last leaf): {(hi-lo)/1e6:.3f} ms; gaps {gap_tot/1e6:.3f} ms")
for k, (n, d) in sorted(cls.items(), key=lambda kv: -kv[1][1]):
    print(f"   {k:8s} n {n:4d} total {d/1e6:7.3f} ms avg {d/n/1e3:7.1f} us")
