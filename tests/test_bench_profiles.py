"""bench.py reads the PMC-derived fields of its roofline object (HBM-side traffic, matrix-pipe utilisation, shader
clock of the dominant launch) from the committed profiles by KERNEL NAME.  A kernel renamed in the library without
a fresh profile collection silently turns those fields into null -- this keeps the two in step (CPU only)."""
import importlib.util
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_dominant_kernel_name_is_in_the_committed_profiles_and_in_the_source():
    bench = load_bench()
    name = "gemm_epi_kernel<double, false, true, 2>"
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'gemm_epi_kernel<{rname}, false, true, 2>' in src            # what bench.py looks up
    hip = open(os.path.join(ROOT, "gaussian_processes_amd", "csrc", "gemm.hip")).read()
    assert re.search(r"gemm_epi_kernel<R, false, true, 2>", hip)          # what the library launches for T
    t = bench.profiled_traffic(name)
    assert t is not None and t["blocks"] == 2080 and 5.0 < t["fetch_corrected"] < 20.0 and 0.1 < t["write"] < 1.0
    u = bench.profiled_mfma_util(name)
    assert u is not None and 0.5 < u["mfma_pipe_utilisation"] <= 1.0 and 1.5 < u["shader_clock_ghz"] < 2.5
    stats = open(bench.latest_profile("r*_kernel_stats.csv")).read()
    assert name in stats


def test_committed_bench_line_carries_the_contract_fields():
    bench = load_bench()
    j = json.load(open(bench.latest_profile("r*_bench.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    r = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert isinstance(r["traffic"], (int, float)) and r["traffic"] > r["algorithmic_bytes_per_launch"]
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] >= 1
    assert j["config"]["workload"].startswith("N=8192 d=256")


def test_every_committed_bench_line_has_a_sane_roofline():
    """A fraction of peak above 1 is not a measurement (round 2 committed 4.5e6 for sizes whose dominant launch did not
    exist): every JSON line under profiles/ of the current round must carry frac in (0, 1] or null with a note, the
    metric label must name the size it was taken at, and PMC-derived fields may only ride on the profiled headline run."""
    import glob
    bench = load_bench()
    tag = os.path.basename(bench.latest_profile("r*_bench.json")).split("_")[0]
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"{tag}_*.json")) + glob.glob(os.path.join(ROOT, "profiles", f"{tag}_*.jsonl")))
    lines = []
    for f in files:
        for raw in open(f):
            raw = raw.strip()
            if raw.startswith("{") and '"roofline"' in raw:
                try:
                    lines.append((os.path.basename(f), json.loads(raw)))
                except json.JSONDecodeError:
                    pass                  # a pretty-printed (multi-line) file: handled below
        if f.endswith(".json"):
            try:
                j = json.load(open(f))
                if isinstance(j, dict) and "roofline" in j:
                    lines.append((os.path.basename(f), j))
            except json.JSONDecodeError:
                pass
    assert len(lines) >= 10, [f for f, _ in lines]
    for name, j in lines:
        r = j["roofline"]
        assert r["frac"] is None or 0.0 < r["frac"] <= 1.0, (name, r["frac"])
        assert (r["frac"] is None) == (r["achieved"] is None), name
        if r["frac"] is None:
            assert r.get("note"), name
        assert 0.0 < r["unit_executed_frac"] <= 1.0, (name, r["unit_executed_frac"])
        cfg = j["config"]
        if "N=" in j["metric"]:
            assert f"N={cfg['N']}" in j["metric"], (name, j["metric"], cfg["N"])
        headline = cfg["N"] == 8192 and cfg["d"] == 256 and j["dtype"] == "f64" and cfg["workload"].startswith("N=8192 d=256 single cell")
        if not headline:
            assert r.get("traffic") is None and r.get("mfma_util") is None and r.get("clock") is None, name
        assert "unit_algorithmic_tflops_equivalent" not in r, name
    sweep = [j for n, j in lines if n == f"{tag}_size_sweep.jsonl"]
    assert [j["config"]["N"] for j in sweep][:7] == [1024, 2048, 4096, 6144, 8192, 12288, 16384]
    assert sweep[0]["roofline"]["frac"] is None and sweep[4]["roofline"]["frac"] is not None
