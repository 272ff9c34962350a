"""Print the GEMM dispatch sequence of a rocprofv3 --pmc run with the counter value per dispatch."""
import csv, glob, sys, os
d = sys.argv[1]
f = max(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
last = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void gpfit::", "")
    key = (name, r["Grid_Size"], r["Counter_Name"])
    v = float(r["Counter_Value"])
    if key != last:
        print()
        print(f"{name[:60]:60s} blocks {int(r['Grid_Size'])//int(r['Workgroup_Size']):5d} {r['Counter_Name']}:", end=" ")
        last = key
    print(f"{v*1024/1e9*(2 if r['Counter_Name']=='FETCH_SIZE' else 1):.2f}", end=" ")
print()
