"""List the launches of one steady-state fit after the join (T onwards) from a rocprofv3 kernel trace."""
import csv, sys
t = list(csv.DictReader(open(sys.argv[1])))
for r in t:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
t.sort(key=lambda r: r['s'])
starts = [i for i, r in enumerate(t) if 'localker_kernel' in r['Kernel_Name']]
fit = t[starts[2]:starts[3]]
k = max(i for i, r in enumerate(fit) if 'chol_leaf' in r['Kernel_Name'])
tot = 0
for r in fit[k + 1:]:
    d = (r['e'] - r['s']) / 1e3
    tot += d
    name = r['Kernel_Name'].split('(')[0].replace('void gpfit::', '')[:58]
    if d > 30: print(f"{d:9.1f} us  blocks {int(r.get('Grid_Size_X', r.get('Grid_Size', 0)))//max(1, int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1)))):5d}  {name}")
print(f"total after the last leaf: {tot/1e3:.3f} ms")
