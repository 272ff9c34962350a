"""Pair the GEMM launch log of one evaluation (GPFIT_GEMM_LOG=k, stderr) with a rocprofv3 kernel trace of the same
single-stream run: per launch shape, duration and executed TFLOP/s; totals by size class.

    GPFIT_GEMM_LOG=3 GPFIT_SIDE_MIN=0 GPFIT_TS_SIDE=0 rocprofv3 --kernel-trace ... -- python scripts/scratch/dev_lockstep.py 2> log
    python scripts/trace_gemm_rates.py <kernel_trace.csv> <log> <k>"""
import collections, csv, re, sys
trace, log, k = sys.argv[1], sys.argv[2], int(sys.argv[3])
launches = []
for line in open(log):
    m = re.search(r"\[gpfit gemm\] M (\d+) N (\d+) K (\d+) atri (\d+) btri (\d+) lower (\d+) nb (\d+) tile (\d+) ak (\d+) bk (\d+) epi (\d+) flops (\S+)", line)
    if m:
        v = list(m.groups())
        launches.append(dict(M=int(v[0]), N=int(v[1]), K=int(v[2]), atri=int(v[3]), btri=int(v[4]), lower=int(v[5]), nb=int(v[6]),
                             tile=int(v[7]), flops=float(v[11]), epi=int(v[10])))
        continue
    # a paired launch (gemm_pair_kernel): the update of a node and the first product of its inverse merge in one kernel
    m = re.search(r"\[gpfit gemm\] pair: M (\d+) N (\d+) K (\d+) lower 1 nb (\d+) \+ M (\d+) N (\d+) K (\d+) btri 1 nb (\d+) tile (\d+) flops (\S+)", line)
    if m:
        v = list(m.groups())
        launches.append(dict(M=int(v[0]), N=int(v[1]), K=int(v[2]), atri=0, btri=0, lower=1, nb=int(v[3]) + int(v[7]), tile=int(v[8]),
                             flops=float(v[9]), epi=0))
t = list(csv.DictReader(open(trace)))
for r in t:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
t.sort(key=lambda r: r['s'])
starts = [i for i, r in enumerate(t) if 'localker_kernel' in r['Kernel_Name']]
# the k-th evaluation of the process = the k-th localker launch that is followed by a gram kernel (fit_eval)
evals = [i for i in starts if any('gram_acos' in x['Kernel_Name'] for x in t[i:i + 8])]
lo = evals[k - 1]; hi = evals[k] if k < len(evals) else len(t)
fit = t[lo:hi]
gemms = [r for r in fit if ('gemm_mfma' in r['Kernel_Name'] or 'gemm_epi' in r['Kernel_Name'] or 'gemm_xcd' in r['Kernel_Name'] or 'gemm_pair' in r['Kernel_Name']
                           or 'gemm_streamk_kernel' in r['Kernel_Name'])]
# a uniform launch with a stream-K tail is two kernels (head + tail): merge a streamk kernel into the preceding head when the log has one entry
out = []
gi = 0
for L in launches:
    if gi >= len(gemms): break
    r = gemms[gi]; dur = r['e'] - r['s']; gi += 1
    if gi < len(gemms) and 'gemm_streamk_kernel' in gemms[gi]['Kernel_Name'] and 'gemm_streamk_kernel' not in r['Kernel_Name'] and L['tile'] == 128 \
            and L['atri'] == 0 and L['btri'] == 0 and L['nb'] == 1:
        ntiles = (L['M'] // 128) * ((L['M'] // 128) + 1) // 2 if L['lower'] else (L['M'] // 128) * (L['N'] // 128)
        if ntiles > 512 and ntiles % 512 and ntiles % 512 < 384:
            dur += gemms[gi]['e'] - gemms[gi]['s']; gi += 1
    out.append((L, dur / 1e3))
print(f"{len(launches)} logged launches, {len(gemms)} gemm kernels in the trace window, paired {len(out)} (consumed {gi})")
cls = collections.defaultdict(lambda: [0, 0.0, 0.0])
for L, us in out:
    size = max(L['M'], L['N'])
    key = ("%5d" % size, L['tile'])
    cls[key][0] += 1; cls[key][1] += us; cls[key][2] += L['flops']
    if us > 100:
        print(f"  M {L['M']:5d} N {L['N']:5d} K {L['K']:5d} tri {L['atri']}{L['btri']} lower {L['lower']} nb {L['nb']:2d} tile {L['tile']:3d} epi {L['epi']}: {us:8.1f} us  {L['flops'] / us / 1e6:6.1f} TF/s")
print("by output size and tile: launches, ms, executed TF/s")
tot_us = tot_fl = 0
for key, (n, us, fl) in sorted(cls.items()):
    print(f"  size {key[0]} tile {key[1]:3d}: n {n:4d}  {us / 1e3:8.3f} ms  {fl / us / 1e6:6.1f} TF/s")
    tot_us += us; tot_fl += fl
print(f"  all: {tot_us / 1e3:.3f} ms, {tot_fl / 1e12:.3f} TFLOP, {tot_fl / tot_us / 1e6:.1f} TF/s")
