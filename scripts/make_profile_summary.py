"""Condense rocprofv3 outputs (gpurun_out/...) into the small tracked files under profiles/.

    python scripts/make_profile_summary.py <round-tag> <kernel-trace-dir> [<pmc-fetch-dir> <pmc-write-dir>]
"""
import collections, csv, glob, json, os, sys

tag, trace_dir = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(root, "profiles")
os.makedirs(out_dir, exist_ok=True)


def one(pattern):
    m = glob.glob(os.path.join(pattern))
    assert m, pattern
    return max(m, key=os.path.getmtime)  # gpurun merges every call's files into the same directory: newest wins


stats = list(csv.DictReader(open(one(os.path.join(trace_dir, "*", "*_kernel_stats.csv")))))
with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "percent"])
    for r in stats:
        w.writerow([r["Name"], r["Calls"], f"{float(r['TotalDurationNs'])/1e6:.3f}", f"{float(r['AverageNs'])/1e3:.2f}",
                    f"{float(r['MinNs'])/1e3:.2f}", f"{float(r['MaxNs'])/1e3:.2f}", r["Percentage"]])

summary = {"source": "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline"}
gemm = [r for r in stats if "gemm_mfma_kernel" in r["Name"] or "gemm_streamk_kernel" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in gemm)
calls = sum(int(r["Calls"]) for r in gemm)
summary["gemm_mfma_kernel+gemm_streamk_kernel"] = {"calls": calls, "total_ms": tot / 1e6, "avg_launch_us": tot / calls / 1e3}
allk = sum(float(r["TotalDurationNs"]) for r in stats)
summary["all_kernels_total_ms"] = allk / 1e6

if len(sys.argv) >= 5:
    def agg(d, cname):
        rows = csv.DictReader(open(one(os.path.join(d, "*", "*_counter_collection.csv"))))
        a = collections.defaultdict(lambda: [0, 0.0, 0.0])
        for r in rows:
            if r["Counter_Name"] != cname:
                continue
            k = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
            a[k][0] += 1
            a[k][1] += float(r["Counter_Value"])
            a[k][2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        return a
    fa, wa = agg(sys.argv[3], "FETCH_SIZE"), agg(sys.argv[4], "WRITE_SIZE")
    rows = []
    for k, (n, v, ms) in sorted(fa.items(), key=lambda kv: -kv[1][1])[:14]:
        wv = wa.get(k, [1, 0.0, 0.0])
        # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a wide coalesced read at half
        # its bytes (MI355X_MICROARCH.md, HBM section) -> doubled for the corrected figure
        rows.append({"kernel": k[0], "blocks": k[1], "launches": n,
                     "fetch_raw_GB_per_launch": v / n * 1024 / 1e9,
                     "fetch_corrected_GB_per_launch": 2 * v / n * 1024 / 1e9,
                     "write_GB_per_launch": wv[1] / max(1, wv[0]) * 1024 / 1e9,
                     "avg_ms_under_pmc": ms / n})
    summary["hbm_traffic_by_kernel"] = rows
json.dump(summary, open(os.path.join(out_dir, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:3000])
