"""Turn the rocprofv3 / bench outputs of scripts/collect_profiles.sh <tag> (gpurun_out/<tag>_*) into the tracked profiles/<tag>_* files."""
import collections, csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "make_profile_summary.py"), tag, os.path.join(G, f"{tag}_kt"),
                       os.path.join(G, f"{tag}_pmcf"), os.path.join(G, f"{tag}_pmcw")], stdout=subprocess.DEVNULL)
shutil.copy(os.path.join(G, f"{tag}_bench_full.json"), os.path.join(P, f"{tag}_bench.json"))
shutil.copy(os.path.join(G, f"{tag}_bench_f32.json"), os.path.join(P, f"{tag}_bench_f32.json"))
for extra in ("bench_mixed.json", "size_sweep.jsonl", "whole_fits.log", "whole_fit_breakdown.json", "active_loop.log"):
    if os.path.exists(os.path.join(G, f"{tag}_{extra}")):
        shutil.copy(os.path.join(G, f"{tag}_{extra}"), os.path.join(P, f"{tag}_{extra}"))
with open(os.path.join(P, f"{tag}_configs.jsonl"), "w") as o:
    o.writelines(l for l in open(os.path.join(G, f"{tag}_configs.jsonl")) if l.startswith("{"))
f = max(glob.glob(os.path.join(G, f"{tag}_pmcm", "*", "*_counter_collection.csv")), key=os.path.getmtime)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
dur = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[k] += 1
        dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
with open(os.path.join(P, f"{tag}_mfma_busy.csv"), "w") as o:
    w = csv.writer(o)
    w.writerow(["kernel", "dispatches", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "busy_per_active",
                "mfma_pipe_utilisation = busy_per_active / 128 (1024 SIMDs / 8 XCD counters)",
                "shader_clock_ghz = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration in ns"])
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
        mf, ga = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), v.get("GRBM_GUI_ACTIVE", 1)
        if mf > 0:
            w.writerow([k, cnt[k], f"{mf:.4e}", f"{ga:.4e}", f"{mf/ga:.2f}", f"{mf/ga/128:.3f}", f"{ga/8/max(dur[k], 1):.3f}"])
for name in (f"{tag}_bench.json", f"{tag}_bench_f32.json"):
    j = json.load(open(os.path.join(P, name))); r = j["roofline"]
    print(name, j["value"], j["unit"], j["ms_per_step"], "ms; roofline", r["achieved"], r["frac"], r["avg_launch_ms"], "family", r["gemm_family"]["tflops"],
          "cpu", j.get("cpu_baseline", {}).get("value"))
