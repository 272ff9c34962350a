"""Headline benchmark: GP fits/sec at N=8192, d=256, fp64 (BASELINE.json configs[2]).

One "step" = one evaluation of the unit of work: spatial metric +
arc-cosine kernel build + Cholesky(K~) + Cholesky(V) + solves + log-marginal + 6 analytic
gradients + lambda moments, on inputs already resident in HBM.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, each rank evaluates its own cell (weak scaling); the shared
stimulus matrix X is generated on rank 0 and broadcast over RCCL once, outside the timed
region; there is no collective on the data path.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from gaussian_processes_amd import synthetic as syn  # noqa: E402
from gaussian_processes_amd.engine import GPFitEngine, fits_flops  # noqa: E402

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X dense fp64 matrix peak (AMD datasheet; rocBLAS dgemm reaches 76.7 on-box)
FP32_MFMA_PEAK_TFLOPS = 157.3  # f32-input MFMA peak (MI355X_MICROARCH.md, chip-level parameters)


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores() -> int:
    """CPU cores this process may actually use: the cgroup quota / affinity mask, not the
    host's core count (a 1-GPU box gives its jobs a 16-core share)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, 32))


def profiled_traffic(kernel_name):
    """HBM bytes per launch of the named kernel, from the committed PMC passes
    (profiles/r*_summary.json: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE of this same
    command, FETCH_SIZE doubled per the gfx950 correction).  PMC collection needs its own
    profiler runs, so this is read from the tracked profile, not measured live; None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))
    if not files:
        return None
    try:
        rows = json.load(open(files[-1])).get("hbm_traffic_by_kernel") or []
        if not rows:
            return None
        match = [x for x in rows if kernel_name in x["kernel"]]
        if not match:
            return None
        r = max(match, key=lambda x: x["fetch_corrected_GB_per_launch"])
        return {"unit": "GB/launch", "kernel": r["kernel"], "blocks": r["blocks"],
                "fetch_corrected": round(r["fetch_corrected_GB_per_launch"], 3),
                "write": round(r["write_GB_per_launch"], 3), "source": os.path.basename(files[-1])}
    except Exception:
        return None


def build_V(X, grid, th0, dev):
    """V = K~(theta0)/2 (SPD by construction).  Setup only, outside the timed
    region; uses the library's own kernel-build entry point."""
    from gaussian_processes_amd import utils as gp
    lower, upper = syn.limits()
    t = {k: torch.tensor(v, dtype=torch.float64) for k, v in th0.items()}
    C, mask = gp.localker(t, upper, lower, grid, grad=False)
    Xm = X[:, mask.to(X.device)] if not bool(mask.all()) else X
    K = gp.acosker(t, Xm, Xm, C=C, dC=None, diag=False)
    return 0.5 * K


def cpu_baseline(n_sample: int, d: int, budget_s: float = 30.0):
    """Reference-formulation closure (oracle.mstep_closure_reference: materialised dK{6},
    eigen-projection, LU inverse, 13+13 GEMM gradient products; torch CPU fp64) timed on the
    host cores of this box on a bounded sample, plus the CPU Cholesky restatement."""
    from oracle import gp_oracle as orc
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: {cores} threads, sample N={n_sample}")
    lower, upper = syn.limits()
    grid = syn.grid_for(d)
    X = torch.from_numpy(syn.stimuli(n_sample, d))
    r_np, m_np = syn.cell_inputs(n_sample)
    r, m = torch.from_numpy(r_np), torch.from_numpy(m_np)
    th0, th1 = syn.theta0(), syn.theta_eval()
    C0, mask0 = orc.spatial_metric(th0, lower, upper, grid)
    K0 = orc.arccos_gram(th0, X[:, mask0], X[:, mask0], C0)
    V = 0.5 * K0
    ev, evec, keep = orc.eigen_basis(K0, 1e-14)  # full-rank family
    B = evec[:, keep]
    m_b, V_b = B.T @ m, B.T @ V @ B
    logA, lam0 = syn.F_PARAMS["logA"], syn.F_PARAMS["lambda0"]
    t0 = time.time()
    loss_ref, _ = orc.mstep_closure_reference(th1, lower, upper, grid, X, X, r, B, m_b, V_b, logA, lam0, tol=1e-14)
    t_first = time.time() - t0
    log(f"cpu_baseline: first reference-formulation eval {t_first:.2f}s")
    reps, times = 0, [t_first]
    while sum(times) + t_first < budget_s and reps < 2:
        t0 = time.time()
        orc.mstep_closure_reference(th1, lower, upper, grid, X, X, r, B, m_b, V_b, logA, lam0, tol=1e-14)
        times.append(time.time() - t0)
        reps += 1
    t_ref = min(times)
    t0 = time.time()
    loss_chol, _ = orc.mstep_closure_cholesky(th1, lower, upper, grid, X, r, m, V, logA, lam0)
    t_chol = time.time() - t0
    return dict(t_ref=t_ref, t_chol=t_chol, cores=cores, n=n_sample, loss_ref=loss_ref, loss_chol=loss_chol,
                repeats=len(times))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--cpu-sample-n", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-grad", action="store_true", help="forward-only unit (not the headline metric)")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64",
                    help="f64 = the reference's precision (headline); f32 = the theta-grid configuration's precision")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    N, d = args.n, args.d
    grid = syn.grid_for(d)
    lower, upper = syn.limits()

    # shared stimuli: generated on rank 0, broadcast over RCCL/xGMI (16 MiB at the headline)
    if rank == 0:
        X = torch.from_numpy(syn.stimuli(N, d)).to(dev)
    else:
        X = torch.empty(N, d, dtype=torch.float64, device=dev)
    if dist is not None:
        dist.broadcast(X, src=0)

    cell = rank  # one independent cell per GPU
    r_np, m_np = syn.cell_inputs(N, cell)
    r, m = torch.from_numpy(r_np).to(dev), torch.from_numpy(m_np).to(dev)
    th0, th1 = syn.theta0(cell), syn.theta_eval(cell)
    eng = GPFitEngine(N, d, device=local_rank)
    if rank == 0:
        log("context ready; building V")
    V = build_V(X, grid, th0, dev)
    torch.cuda.synchronize(dev)
    if rank == 0:
        log("inputs resident; warm-up")
    logA, lam0 = syn.F_PARAMS["logA"], syn.F_PARAMS["lambda0"]
    want_grad = not args.no_grad
    if args.dtype == "f32":
        X, r, m, V = X.float(), r.float(), m.float(), V.float()
    peak = FP64_MFMA_PEAK_TFLOPS if args.dtype == "f64" else FP32_MFMA_PEAK_TFLOPS

    def step():
        return eng.fit_eval(th1, lower, upper, grid, X, r, m, V, logA, lam0, want_grad=want_grad,
                            want_vectors=False)

    for _ in range(args.warmup):
        res = step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert math.isfinite(res["loss"]), "benchmark evaluation produced a non-finite loss"

    if rank == 0:
        log(f"timed region done: {elapsed / args.steps * 1e3:.2f} ms/step (host enqueue {eng.last_enqueue_ms():.2f} ms)")
        fits_per_s = world * args.steps / elapsed
        F = fits_flops(N, d)
        # ---- roofline of the dominant kernel (fp64 MFMA GEMM family), HIP events per launch
        eng.set_profile(True)
        step()
        prof = eng.get_profile()
        eng.set_profile(False)
        gemm_tflops = prof["gemm_flops"] / (prof["gemm_ms"] * 1e-3) / 1e12 if prof["gemm_ms"] > 0 else 0.0
        # dominant kernel = the single largest launch: T = L^-1 L_V (both operands lower triangular,
        # lower output; algorithmic flops N^3/3), the stream-K kernel
        # gemm_streamk_kernel<R,false,true>: one call per fit, so its rocprofv3 kernel_stats row is
        # its average (the event bracket here also covers the ~2 % fix-up kernel behind it).
        npad = -(-N // 128) * 128
        dom_flops = float(npad) ** 3 / 3.0
        dom_tflops = dom_flops / max(prof["largest_gemm_ms"], 1e-9) / 1e9
        dom_name = "gemm_streamk_kernel<%s, false, true>" % ("double" if args.dtype == "f64" else "float")
        roofline = {
            "bound": "mfma",
            "kernel": dom_name + " (T = L^-1 L_V, N^3/3 flops, 1 launch/fit; %s)"
                      % ("v_mfma_f64_16x16x4_f64" if args.dtype == "f64" else "v_mfma_f32_16x16x4_f32"),
            "achieved": round(dom_tflops, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(dom_tflops / peak, 4), "traffic": profiled_traffic(dom_name) if args.dtype == "f64" else None,
            "launches_per_fit": 1, "avg_launch_ms": round(prof["largest_gemm_ms"], 4),
            "algorithmic_flops_per_launch": dom_flops,
            "gemm_family": {"what": "all 128-tile GEMM/SYRK/TRSM/TRTRI launches (gemm_mfma_kernel<..,128> + gemm_streamk_kernel), executed flops",
                            "launches_per_fit": prof["gemm_launches"], "tflops": round(gemm_tflops, 2),
                            "frac": round(gemm_tflops / peak, 4),
                            "avg_launch_ms": round(prof["gemm_ms"] / max(1, prof["gemm_launches"]), 4)},
            "flops_executed_per_fit": prof["gemm_flops"] + prof["small_gemm_flops"] + prof["gram_flops"],
            "gemm_ms_per_fit": round(prof["gemm_ms"], 3), "leaf_ms_per_fit": round(prof["leaf_ms"], 3),
            "small_tile_gemm": {"launches_per_fit": prof["small_gemm_launches"], "ms_per_fit": round(prof["small_gemm_ms"], 3),
                                "tflops": round(prof["small_gemm_flops"] / max(prof["small_gemm_ms"], 1e-9) / 1e9, 2)},
            "gram_ms_per_fit": round(prof["gram_ms"], 3),
            "unit_algorithmic_flops": F,
            "unit_achieved_tflops": round(F * fits_per_s / world / 1e12, 2),
            "unit_frac_of_peak": round(F * fits_per_s / world / 1e12 / peak, 4),
        }
        out = {
            "metric": "GP fits/sec (kernel+chol+solve+grad loglik) at N=8192 d=256" + ("" if args.dtype == "f64" else " [fp32 instance]"),
            "value": round(fits_per_s, 4), "unit": "fits/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"N={N} d={d} single cell {'fp64' if args.dtype == 'f64' else 'fp32'}, one M-step closure evaluation with 6 gradients "
                                   "(BASELINE configs[2], headline)" if want_grad else f"N={N} d={d} forward only",
                       "N": N, "d": d, "cells_per_gpu": 1, "parallelism": f"independent cells x{world}, X broadcast once over RCCL"},
            "loss": res["loss"],
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline and args.dtype == "f64":
            cb = cpu_baseline(args.cpu_sample_n, d)
            scale = (N / cb["n"]) ** 3
            out["cpu_baseline"] = {
                "value": round(1.0 / (cb["t_ref"] * scale), 6), "unit": "fits/s", "cores": cb["cores"], "kind": "port",
                "sample": f"reference-formulation closure (oracle.mstep_closure_reference, torch CPU fp64) at N={cb['n']} "
                          f"d={d}: {cb['t_ref']:.2f} s/eval (best of {cb['repeats']}), scaled x{scale:.0f} (N^3) to N={N}",
                "sample_seconds_per_eval": round(cb["t_ref"], 3),
                "cholesky_port_seconds_per_eval": round(cb["t_chol"], 3),
                "gpu_vs_cpu": round(fits_per_s * cb["t_ref"] * scale, 1),
            }
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
