"""Benchmark of the GP fit hot path on MI355X (BASELINE.json): GP fits/sec at N=8192, d=256, fp64.

One "step" = one pass of the hot path over one batch of synthetic input, inputs already resident
in HBM.  The unit of work is one evaluation of the M-step closure:
spatial metric + arc-cosine kernel build + Cholesky(K~) + Cholesky(V) + solves + log-marginal + six
analytic gradients.

    python bench.py --gpus 1 --steps 10 --warmup 2                     # headline (BASELINE configs[2])
    python bench.py --config n4096                                      # configs[1]: N=4096, d=128
    python bench.py --config cells64                                    # configs[3]: 64 cells x N=4096, sharded
    python bench.py --config thetagrid                                  # configs[4]: 512 theta x N=8192, fp32
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W [--config ...]

N > 1: one process per GPU over RCCL.  headline / n4096: every rank evaluates its own cell (weak
scaling); cells64 / thetagrid: the 64 cells / 512 theta points are sharded cyclically over the ranks
(strong scaling: one step = one pass over all units).  The shared stimulus matrix X is generated on
rank 0 and broadcast once outside the timed region (thetagrid also broadcasts r, m, V:
multi.broadcast_state); there is no collective on the data path.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from gaussian_processes_amd import multi, synthetic as syn  # noqa: E402
from gaussian_processes_amd.engine import (GPFitEngine, fit_eval_group, fit_eval_group_begin, fit_eval_group_finish,  # noqa: E402
                                           fits_flops)

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X dense fp64 matrix peak (AMD datasheet; rocBLAS dgemm reaches 76.7 on-box)
NOMINAL_GHZ = 2.4                 # MI355X_MICROARCH.md: the clock the peak figures are quoted at
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


def latest_profile(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return files[-1] if files else None


def profiled_traffic(kernel_name, blocks=None):
    """HBM-side bytes per launch of the named kernel, from the committed PMC passes
    (profiles/r*_summary.json: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE of this same
    command, FETCH_SIZE doubled per the gfx950 correction).  PMC collection needs its own
    profiler runs, so this is read from the tracked profile, not measured live; None if absent."""
    f = latest_profile("r*_summary.json")
    if not f:
        return None
    try:
        rows = json.load(open(f)).get("hbm_traffic_by_kernel") or []
        match = [x for x in rows if kernel_name in x["kernel"] and (blocks is None or x["blocks"] == blocks)]
        if not match:
            return None
        r = max(match, key=lambda x: x["fetch_corrected_GB_per_launch"])
        return {"unit": "GB/launch", "kernel": r["kernel"], "blocks": r["blocks"],
                "fetch_corrected": round(r["fetch_corrected_GB_per_launch"], 3),
                "write": round(r["write_GB_per_launch"], 3), "source": os.path.basename(f)}
    except Exception:
        return None


def profiled_mfma_util(kernel_name):
    """Matrix-pipe utilisation of the named kernel from the committed PMC pass
    (profiles/r*_mfma_busy.csv: SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE / 128); None if absent."""
    f = latest_profile("r*_mfma_busy.csv")
    if not f:
        return None
    try:
        for r in csv.reader(open(f)):
            if r and kernel_name in r[0]:
                out = {"mfma_pipe_utilisation": float(r[5]), "dispatches": int(r[1]), "source": os.path.basename(f)}
                if len(r) > 6:      # GRBM_GUI_ACTIVE / 8 XCDs / duration: the shader clock the launch really ran at
                    out["shader_clock_ghz"] = float(r[6])
                return out
    except Exception:
        pass
    return None


PROFILED_EVALUATIONS = 5   # evaluations behind the live per-launch figures of `roofline` (HIP events around every GEMM launch)


def clamp_groups(args, n_local):
    """A rank never allocates contexts it cannot fill: the group size is clamped to the rank's local unit count and the
    number of engine sets to the number of groups that leaves (at --gpus 8, cells64 gives every rank 8 cells: ONE group
    of 8 on 8 contexts, not a half-full group of 16 on 32 contexts = 56 GB).  Every rank holds the same count up to one
    unit (cyclic partition), and the clamp only ever shrinks, so results do not change: a unit's numbers do not depend
    on the group it is evaluated in."""
    if args.group > 0:
        args.group = max(1, min(args.group, n_local))
        args.sets = max(1, min(args.sets, -(-max(1, n_local) // args.group)))


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


def in_flight_rate(eng, N, d, X, grid, dev, tdt, lower, upper, logA, lam0, want_grad, gprec, local_rank, depth=3, cells=6, rounds=3):
    """fits/s with `depth` independent cells of the headline size in flight (multi.evaluate_units_pipelined)."""
    inputs = []
    for c in range(cells):
        rc, mc = syn.cell_inputs(N, c)
        inputs.append((torch.from_numpy(rc).to(dev).to(tdt), torch.from_numpy(mc).to(dev).to(tdt),
                       build_V(X, grid, syn.theta0(c), dev).to(tdt), syn.theta_eval(c)))
    Xd = X.to(tdt)
    engs = [eng] + [GPFitEngine(N, d, device=local_rank) for _ in range(depth - 1)]
    streams = [torch.cuda.Stream(device=dev) for _ in engs]

    def submit(u, slot):
        rc, mc, Vc, thc = inputs[u % cells]
        with torch.cuda.stream(streams[slot]):
            return engs[slot].fit_eval_async(thc, lower, upper, grid, Xd, rc, mc, Vc, logA, lam0, want_grad=want_grad,
                                             want_vectors=False, grad_precision=gprec)

    def collect(t, slot):
        o = engs[slot].fit_eval_finish(t)
        return [o["loss"]] + [o["grad"][k] for k in syn.THETA_KEYS]

    units = list(range(cells * rounds))
    multi.evaluate_units_pipelined(units[:cells], submit, collect, dev, depth)          # warm-up
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    table = multi.evaluate_units_pipelined(units, submit, collect, dev, depth)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    for e in engs[1:]:
        e.close()
    assert bool(torch.isfinite(table).all())
    return {"fits_per_s": round(len(units) / dt, 3), "ms_per_fit": round(dt / len(units) * 1e3, 3), "in_flight": depth,
            "what": f"{len(units)} evaluations of {cells} independent cells of the headline size, {depth} in flight on {depth} contexts"}


def cpu_reference_eval(n, d, repeats, do_cholesky=True):
    """Reference-formulation closure (oracle.mstep_closure_reference: materialised dK{6},
    eigen-projection, LU inverse, 13+13 GEMM gradient products; torch CPU fp64) on the host cores of
    this box: one warm-up + `repeats` timed evaluations; optionally the CPU Cholesky restatement."""
    from oracle import gp_oracle as orc
    lower, upper = syn.limits()
    grid = syn.grid_for(d)
    X = torch.from_numpy(syn.stimuli(n, d))
    r_np, m_np = syn.cell_inputs(n)
    r, m = torch.from_numpy(r_np), torch.from_numpy(m_np)
    th0, th1 = syn.theta0(), syn.theta_eval()
    C0, mask0 = orc.spatial_metric(th0, lower, upper, grid)
    K0 = orc.arccos_gram(th0, X[:, mask0], X[:, mask0], C0)
    V = 0.5 * K0
    ev, evec, keep = orc.eigen_basis(K0, 1e-14)  # full-rank family
    B = evec[:, keep]
    m_b, V_b = B.T @ m, B.T @ V @ B
    del K0, ev, evec
    logA, lam0 = syn.F_PARAMS["logA"], syn.F_PARAMS["lambda0"]
    times = []
    for it in range(repeats + (1 if repeats > 1 else 0)):
        t0 = time.time()
        loss_ref, _ = orc.mstep_closure_reference(th1, lower, upper, grid, X, X, r, B, m_b, V_b, logA, lam0, tol=1e-14)
        times.append(time.time() - t0)
        log(f"cpu_baseline: N={n} reference-formulation eval {it}: {times[-1]:.2f} s")
    timed = times[1:] if repeats > 1 else times
    out = {"n": n, "t_ref": float(np.median(timed)), "t_ref_all": [round(t, 3) for t in timed], "loss_ref": loss_ref}
    if do_cholesky:
        t0 = time.time()
        loss_chol, _ = orc.mstep_closure_cholesky(th1, lower, upper, grid, X, r, m, V, logA, lam0)
        out["t_chol"] = time.time() - t0
        out["loss_chol"] = loss_chol
    return out


def cpu_baseline(n_full, d, n_sample, quick):
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: {cores} threads")
    small = cpu_reference_eval(n_sample, d, repeats=3)
    full = None
    if not quick and n_full > n_sample:
        full = cpu_reference_eval(n_full, d, repeats=1)
    return cores, small, full


def projected_report(args, eng, step, res, N, d, ntilde, n_kept, ms_per_step, units_per_s, world, X, Xt, r, Bq, m_b, V_b, th1,
                     lower, upper, grid, logA, lam0):
    """JSON line of --config trunc / sparse: the B-projected closure (utils.py:2047-2099) at the reference's default
    EIGVAL_TOL.  Executed flops and per-family times from HIP events around every GEMM / Gram launch of one more
    evaluation; roofline of the largest 128-tile launch (K_b = K B or the lift W = (.) B^T, 2 N n_t n flops each)."""
    eng.set_profile(1)
    step()
    prof = eng.get_profile()
    eng.set_profile(0)
    peak = FP64_MFMA_PEAK_TFLOPS
    executed = prof["gemm_flops"] + prof["small_gemm_flops"] + prof["gram_flops"]
    big_ms, big_fl = prof["largest_gemm_ms"], prof["largest_gemm_flops"]
    gemm_tflops = prof["gemm_flops"] / (prof["gemm_ms"] * 1e-3) / 1e12 if prof["gemm_ms"] > 0 else 0.0
    nb = -(-n_kept // 128) * 128
    # algorithmic flops of the projected closure as the reference writes it, with the gradient in adjoint form (no
    # dK materialised): kernel build 2 n~^2 d (+ 2 n_t n~ d rectangular), projections K B and B^T (K B):
    # 2 n_t n~ n + 2 n~ n^2 (+ 2 n~^2 n when the two kernels differ), n x n algebra O(n^3), N x n products with n x n
    # matrices (a V_b, G_a K^-1, B G: 3 x 2 n_t n^2), the lift W = (.) B^T 2 n_t n~ n (twice when sparse), pull-back 2 n~^2 d
    nt_ = N
    alg = (2.0 * ntilde * ntilde * d + 2.0 * nt_ * ntilde * n_kept + 2.0 * ntilde * n_kept ** 2 + 6.0 * nt_ * n_kept ** 2
           + 2.0 * nt_ * ntilde * n_kept + 2.0 * ntilde * ntilde * d + (14.0 / 3.0) * n_kept ** 3)
    if ntilde != nt_:
        alg += 2.0 * nt_ * ntilde * d + 2.0 * ntilde * ntilde * n_kept + 2.0 * ntilde * ntilde * n_kept + 2.0 * nt_ * ntilde * d
    roofline = {
        "bound": "mfma",
        "kernel": "gemm_mfma_kernel<double, ..., 128, 2>: the largest 128-tile launch of the closure (K_b = K B / the lift W = (.) B^T: "
                  "2 n_t n~ nb flops, nb = n_kept rounded up to 128); v_mfma_f64_16x16x4_f64",
        "achieved": round(big_fl / big_ms / 1e9, 2) if big_ms > 0 else None, "peak": peak, "unit": "TFLOP/s",
        "frac": round(big_fl / big_ms / 1e9 / peak, 4) if big_ms > 0 else None,
        "traffic": None,
        "avg_launch_ms": round(big_ms, 4) if big_ms > 0 else None, "executed_flops_per_launch": big_fl,
        "gemm_family": {"what": "all 128-tile GEMM launches, executed flops", "launches_per_fit": prof["gemm_launches"],
                        "tflops": round(gemm_tflops, 2), "frac": round(gemm_tflops / peak, 4), "ms_per_fit": round(prof["gemm_ms"], 3)},
        "small_tile_gemm": {"launches_per_fit": prof["small_gemm_launches"], "ms_per_fit": round(prof["small_gemm_ms"], 3),
                            "tflops": round(prof["small_gemm_flops"] / max(prof["small_gemm_ms"], 1e-9) / 1e9, 2)},
        "leaf_launches_per_fit": prof["leaf_launches"], "leaf_ms_per_fit": round(prof["leaf_ms"], 3),
        "gram_ms_per_fit": round(prof["gram_ms"], 3),
        "flops_executed_per_fit": executed, "unit_ms": round(ms_per_step, 3),
        "unit_executed_tflops": round(executed / (ms_per_step * 1e-3) / 1e12, 2),
        "unit_executed_frac": round(executed / (ms_per_step * 1e-3) / 1e12 / peak, 4),
        "unit_algorithmic_flops": alg,
        # compulsory HBM traffic of the unit: K~ written and read once, cos(delta) likewise, W written and read once, A_w
        # written and read once (n~^2 doubles each), the N x nb panels a few times -- against the time at 8 TB/s
        "hbm_floor_ms": round((8.0 * ntilde * ntilde * 8 + 10.0 * N * nb * 8) / 8e12 * 1e3, 3),
    }
    out = {
        "metric": ("GP fits/sec, B-projected (truncated-rank) M-step closure at the reference's default EIGVAL_TOL, "
                   f"N={N} d={d}, {n_kept} of {ntilde} eigen-directions kept") if ntilde == N else
                  (f"GP fits/sec, sparse M-step closure n_t={N} n_tilde={ntilde} d={d} at the reference's default EIGVAL_TOL, "
                   f"{n_kept} eigen-directions kept"),
        "value": round(units_per_s, 4), "unit": "fits/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"{args.config}: n_t={N} n_tilde={ntilde} d={d} f64, one evaluation of the B-projected closure "
                               f"(utils.py:2047-2099) with 6 gradients, basis B[{ntilde} x {n_kept}] fixed (as during an M-step)",
                   "N": N, "d": d, "n_tilde": ntilde, "n_kept": n_kept, "units_per_step": world,
                   "parallelism": f"independent units over {world} GPU(s), one process per GPU, no data-path collective"},
        "loss": res["loss"], "roofline": roofline,
    }
    if world == 1 and not args.no_cpu_baseline:
        from oracle import gp_oracle as orc
        cores = host_cores()
        torch.set_num_threads(cores)
        cpu = [t.detach().cpu() for t in (X, Xt, r, Bq, m_b, V_b)]
        t0 = time.time()
        loss_ref, grad_ref = orc.mstep_closure_reference(th1, lower, upper, grid, cpu[0], cpu[1], cpu[2], cpu[3], cpu[4], cpu[5], logA, lam0,
                                                         tol=1e-4)
        t_ref = time.time() - t0
        g_ref = np.array([grad_ref[k] for k in syn.THETA_KEYS]); g_gpu = np.array([res["grad"][k] for k in syn.THETA_KEYS])
        out["cpu_baseline"] = {
            "value": round(1.0 / t_ref, 6), "unit": "fits/s", "cores": cores, "kind": "port",
            "sample": f"ONE evaluation of the reference-formulation closure (oracle.mstep_closure_reference, same B, m_b, V_b) at "
                      f"n_t={N} n_tilde={ntilde} d={d}: {t_ref:.1f} s",
            "gpu_vs_cpu": round(units_per_s / world * t_ref, 1),
            "loss_rel_dev_gpu_vs_cpu": abs(res["loss"] - loss_ref) / abs(loss_ref),
            "grad_dev_rel_to_largest_component": float(np.abs(g_ref - g_gpu).max() / np.abs(g_ref).max()),
        }
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=["headline", "n4096", "cells64", "thetagrid", "trunc", "sparse"], default="headline",
                    help="BASELINE.json configs[2] (default), [1], [3], [4]; trunc = the headline inputs at the reference's default "
                         "EIGVAL_TOL (the B-projected closure the reference actually runs there, utils.py:2047-2099); sparse = n_t = N, "
                         "n_tilde = N / 4 inducing stimuli (utils.py:1677, 1693)")
    ap.add_argument("--n", type=int, default=0, help="override N (headline / n4096 only)")
    ap.add_argument("--d", type=int, default=0, help="override d")
    ap.add_argument("--cells", type=int, default=64)
    ap.add_argument("--grid-points", type=int, default=512)
    ap.add_argument("--group", type=int, default=None,
                    help="independent units per grouped call (gpfit_fit_eval_batch: their factorisations in lock step, "
                         "batched launches); default 16 for cells64 and thetagrid; 0 = the pipelined driver of --depth")
    ap.add_argument("--sets", type=int, default=None,
                    help="sets of --group engines for the grouped configurations: with 2 the next group is enqueued before the "
                         "previous one is collected (the host's share of a group runs beside the GPU's); 1 = one group at a time. "
                         "Default 2 for cells64, 1 for thetagrid (a second set of 8192-sized contexts is 60 GB for a host share "
                         "of 2 %% of a group)")
    ap.add_argument("--depth", type=int, default=None,
                    help="with --group 0: independent units kept in flight per GPU on as many contexts (default 3)")
    ap.add_argument("--cpu-sample-n", type=int, default=4096)
    ap.add_argument("--cpu-quick", action="store_true", help="skip the real N=8192 CPU evaluation (about 90 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-in-flight", action="store_true", help="skip the informational leg with independent cells in flight")
    ap.add_argument("--no-grad", action="store_true", help="forward-only unit (not the headline metric)")
    ap.add_argument("--no-accuracy-check", action="store_true",
                    help="thetagrid in a reduced-precision mode: skip the fp64 sweep of the same lattice after the timed region")
    ap.add_argument("--dtype", choices=["f64", "f32", "mixed"], default=None,
                    help="f64 = the reference's precision; mixed = fp64 factorisations and loss, fp32 gradient products "
                         "(the theta-grid configuration's default: the all-fp32 instance misses the 1e-5 bar on part of "
                         "the lattice); f32 = every matrix in fp32")
    args = ap.parse_args()
    args.depth_given = args.depth is not None
    if args.depth is None:
        args.depth = 3   # measured optimum for both configs this round (cells64: 148 / 183 / 168 / 166 / 168 cells/s at 2..6)
    if args.group is None:
        args.group = {"cells64": 16, "thetagrid": 16}.get(args.config, 0)   # thetagrid: 16 contexts of N = 8192 are 118 GB
    if args.sets is None:
        args.sets = 2 if args.config == "cells64" else 1
    args.group_asked, args.sets_asked = args.group, args.sets
    dtype_name = args.dtype or ("mixed" if args.config == "thetagrid" else "f64")

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

    defaults = {"headline": (8192, 256), "n4096": (4096, 128), "cells64": (4096, 128), "thetagrid": (8192, 256),
                "trunc": (8192, 256), "sparse": (8192, 256)}
    N, d = defaults[args.config]
    N, d = args.n or N, args.d or d
    grid = syn.grid_for(d)
    lower, upper = syn.limits()
    tdt = torch.float32 if dtype_name == "f32" else torch.float64
    gprec = "f32" if dtype_name == "mixed" else "native"
    peak = FP32_MFMA_PEAK_TFLOPS if dtype_name == "f32" else FP64_MFMA_PEAK_TFLOPS
    logA, lam0 = syn.F_PARAMS["logA"], syn.F_PARAMS["lambda0"]
    want_grad = not args.no_grad

    # shared stimuli: generated on rank 0, broadcast over RCCL/xGMI (16 MiB at the headline)
    X = multi.broadcast_stimuli(torch.from_numpy(syn.stimuli(N, d)) if rank == 0 else None, (N, d), dev).to(dev)

    def sync_barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    eng = GPFitEngine(N, d, device=local_rank)
    extra_engines = []
    if args.config in ("headline", "n4096"):
        cell = rank  # one independent cell per GPU (weak scaling)
        r_np, m_np = syn.cell_inputs(N, cell)
        r, m = torch.from_numpy(r_np).to(dev), torch.from_numpy(m_np).to(dev)
        th1 = syn.theta_eval(cell)
        V = build_V(X, grid, syn.theta0(cell), dev)
        Xd, rd, md, Vd = (t.to(tdt) for t in (X, r, m, V))
        units_per_step, unit_name, scaling = world, "fits", "weak"

        def step():
            return eng.fit_eval(th1, lower, upper, grid, Xd, rd, md, Vd, logA, lam0, want_grad=want_grad,
                                want_vectors=False, grad_precision=gprec)
    elif args.config in ("trunc", "sparse"):
        # The regime the reference runs at its default tolerance (utils.py:39, 1683): K~ of these inputs keeps ~520 of
        # its 8192 eigenvalues, every M-step closure is the B-projected one (utils.py:2047-2099) -- here ONE call of
        # gpfit_fit_eval_projected / gpfit_fit_eval_sparse through the drop-in module, exactly the call varGP makes.
        from gaussian_processes_amd import utils as gp
        cell = rank
        ntilde = N if args.config == "trunc" else N // 4
        r_np, m_np = syn.cell_inputs(N, cell)
        r = torch.from_numpy(r_np).to(dev)
        m = torch.from_numpy(m_np).to(dev)[:ntilde].contiguous()
        th0t = {k: torch.tensor(v, dtype=torch.float64) for k, v in syn.theta0(cell).items()}
        th1 = syn.theta_eval(cell)
        Xt = X if ntilde == N else X[:ntilde].contiguous()
        C0, mask0 = gp.localker(th0t, upper, lower, grid, grad=False)
        Xt_m = Xt[:, mask0.to(dev)].contiguous() if not bool(mask0.all()) else Xt
        K0 = gp.acosker(th0t, Xt_m, Xt_m, C=C0, dC=None, diag=False)
        _, Bq, Ktb0, _ = gp._stabilised_basis(K0)          # default EIGVAL_TOL: the reference's truncation rule
        n_kept = int(Bq.shape[1])
        assert n_kept < ntilde, "these inputs were expected to truncate at the default tolerance"
        m_b = (Bq.T @ m).contiguous()
        V_b = (0.5 * Ktb0).contiguous()                     # V = K~(theta0) / 2  ->  V_b = B^T V B
        del K0
        f_par = {"logA": torch.tensor(logA, dtype=torch.float64), "lambda0": torch.tensor(lam0, dtype=torch.float64)}
        lims = (lower, upper)
        units_per_step, unit_name, scaling = world, "fits", "weak"
        eng.close()
        eng = gp.get_engine(N, d, d)                        # the context the drop-in module evaluates on

        def step():
            if args.config == "trunc":
                loss, grad = gp._closure_projected(th1, lims, grid, X, r, Bq, m_b, V_b, f_par)
            else:
                loss, grad = gp._closure_sparse(th1, lims, grid, X, Xt, r, Bq, m_b, V_b, f_par)
            return {"loss": loss, "grad": grad}
    elif args.config == "cells64":
        cells = args.cells
        mine = multi.partition(cells, world, rank)
        inputs = {}
        for c in mine:  # per-cell inputs built outside the timed region
            rc, mc = syn.cell_inputs(N, c)
            inputs[c] = (torch.from_numpy(rc).to(dev), torch.from_numpy(mc).to(dev), build_V(X, grid, syn.theta0(c), dev),
                         syn.theta_eval(c))
        clamp_groups(args, len(mine))
        n_eng = max(1, args.group) * max(1, args.sets) if args.group > 0 else max(1, args.depth)
        engs = [eng] + [GPFitEngine(N, d, device=local_rank) for _ in range(n_eng - 1)]
        extra_engines = engs[1:]
        streams = [torch.cuda.Stream(device=dev) for _ in engs]

        def rows_of(res):
            return [[o["loss"]] + [o["grad"][k] for k in syn.THETA_KEYS] for o in res]

        def begin_cells(cs, slot):   # one set of engines per slot: the next group is enqueued before this one is collected
            sel = [inputs[c] for c in cs]
            return fit_eval_group_begin(engs[slot * args.group:(slot + 1) * args.group], [t[3] for t in sel], lower, upper, grid, X,
                                        [t[0] for t in sel], [t[1] for t in sel], [t[2] for t in sel], logA, lam0, want_grad=want_grad)

        def finish_cells(handle, slot):
            return rows_of(fit_eval_group_finish(handle))

        def group_cells(cs):
            return finish_cells(begin_cells(cs, 0), 0)

        def submit(c, slot):
            rc, mc, Vc, thc = inputs[c]
            with torch.cuda.stream(streams[slot]):
                return engs[slot].fit_eval_async(thc, lower, upper, grid, X, rc, mc, Vc, logA, lam0, want_grad=want_grad,
                                                 want_vectors=False)

        def collect(ticket, slot):
            o = engs[slot].fit_eval_finish(ticket)
            return [o["loss"]] + [o["grad"][k] for k in syn.THETA_KEYS]

        units_per_step, unit_name, scaling = cells, "cells", "strong"
        table = [None]

        def step():
            if args.group > 0:
                table[0] = multi.run_sharded(cells, None, dev, group_fn=group_cells, group=args.group, begin_fn=begin_cells,
                                             finish_fn=finish_cells, sets=args.sets)
            else:
                table[0] = multi.run_sharded(cells, None, dev, submit_fn=submit, collect_fn=collect, depth=len(engs))
            return {"loss": float(table[0][0, 0])}
    else:  # thetagrid: one cell, 512 theta points, (r, m, V) shared -> broadcast once, V factor reused
        npts = args.grid_points
        points = syn.theta_grid(8)[:npts]
        if rank == 0:
            r_np, m_np = syn.cell_inputs(N, 0)
            r0, m0, V0 = torch.from_numpy(r_np).to(dev), torch.from_numpy(m_np).to(dev), build_V(X, grid, syn.theta0(), dev)
        else:
            r0 = m0 = V0 = None
        rd, md, Vd = multi.broadcast_state(r0, m0, V0, N, dev, dtype=tdt)
        Vd = Vd.contiguous()
        Xd = X.to(tdt)
        if tdt != torch.float64 and not args.no_accuracy_check:
            # the all-fp32 instance is checked against fp64 evaluations of the same points after the timed region
            r64, m64, V64 = multi.broadcast_state(r0, m0, V0, N, dev, dtype=torch.float64)
            V64 = V64.contiguous()
        else:
            r64, m64, V64 = rd, md, Vd
        del r0, m0, V0
        mine = multi.partition(npts, world, rank)
        units_per_step, unit_name, scaling = npts, "theta-points", "strong"
        first = [True]

        # points are independent: --depth of them are kept in flight on as many contexts / streams, so that one
        # point's latency-bound Cholesky chain runs beside another's large gradient products (each context
        # factors V once and then reuses its own copy of the factor)
        tdepth = max(1, args.depth)
        clamp_groups(args, len(mine))
        n_eng = max(1, args.group) * max(1, args.sets) if args.group > 0 else tdepth
        engs = [eng] + [GPFitEngine(N, d, device=local_rank) for _ in range(n_eng - 1)]
        extra_engines = engs[1:]
        streams = [torch.cuda.Stream(device=dev) for _ in engs]
        fresh = [True] * len(engs)
        set_calls = [0] * max(1, args.sets)

        def rows_of(res):
            return [[o["loss"]] + [o["grad"][k] for k in syn.THETA_KEYS] for o in res]

        def begin_points(us, slot):
            # every context factors V in its first group and reuses its own copy of the factor afterwards
            h = fit_eval_group_begin(engs[slot * args.group:(slot + 1) * args.group], [points[u] for u in us], lower, upper, grid, Xd,
                                     rd, md, Vd, logA, lam0, want_grad=want_grad, reuse_V=set_calls[slot] > 0, grad_precision=gprec)
            set_calls[slot] += 1
            return h

        def finish_points(handle, slot):
            return rows_of(fit_eval_group_finish(handle))

        def group_points(us):
            return finish_points(begin_points(us, 0), 0)

        def eval_point(u):
            o = eng.fit_eval(points[u], lower, upper, grid, Xd, rd, md, Vd, logA, lam0, want_grad=want_grad,
                             want_vectors=False, reuse_V=not first[0], grad_precision=gprec)
            first[0] = False
            return [o["loss"]] + [o["grad"][k] for k in syn.THETA_KEYS]

        def submit_point(u, slot):
            with torch.cuda.stream(streams[slot]):
                t = engs[slot].fit_eval_async(points[u], lower, upper, grid, Xd, rd, md, Vd, logA, lam0, want_grad=want_grad,
                                              want_vectors=False, reuse_V=not fresh[slot], grad_precision=gprec)
            fresh[slot] = False
            return t

        def collect_point(ticket, slot):
            o = engs[slot].fit_eval_finish(ticket)
            return [o["loss"]] + [o["grad"][k] for k in syn.THETA_KEYS]

        def step():
            if args.group > 0:
                t = multi.run_sharded(npts, None, dev, group_fn=group_points, group=args.group, begin_fn=begin_points,
                                      finish_fn=finish_points, sets=args.sets)
            elif tdepth == 1:
                t = multi.run_sharded(npts, eval_point, dev)
            else:
                t = multi.run_sharded(npts, None, dev, submit_fn=submit_point, collect_fn=collect_point, depth=tdepth)
            last_table[0] = t
            return {"loss": float(t[0, 0])}

        last_table = [None]

        def reference_sweep():
            """The same lattice in fp64 (the library's reference-precision instance, itself oracle-checked at this size:
            tests/test_gpu_parity.py), grouped, outside the timed region: the yardstick of the reduced-precision modes."""
            calls = [0]

            def group64(us):
                res = fit_eval_group(engs[:len(us)], [points[u] for u in us], lower, upper, grid, X, r64, m64, V64, logA, lam0,
                                     want_grad=want_grad, reuse_V=calls[0] > 0)
                calls[0] += 1
                return [[o["loss"]] + [o["grad"][k] for k in syn.THETA_KEYS] for o in res]
            g = args.group if args.group > 0 else min(8, len(engs))
            return multi.run_sharded(npts, None, dev, group_fn=group64, group=min(g, len(engs)))

    if rank == 0:
        log(f"config {args.config}: inputs resident; warm-up")
    for _ in range(args.warmup):
        res = step()
    sync_barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    sync_barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert math.isfinite(res["loss"]), "benchmark evaluation produced a non-finite loss"
    accuracy = None
    if args.config == "thetagrid" and dtype_name != "f64" and not args.no_accuracy_check:
        # every lattice point of the timed sweep against its fp64 evaluation (north star: 1e-5 relative on the log
        # marginal likelihood); all ranks hold the gathered tables
        ref = reference_sweep()
        got = last_table[0]
        dev_loss = ((got[:, 0] - ref[:, 0]).abs() / ref[:, 0].abs())
        gscale = ref[:, 1:].abs().max(dim=1).values
        dev_grad = ((got[:, 1:] - ref[:, 1:]).abs().max(dim=1).values / gscale)
        accuracy = {"points_compared": int(got.shape[0]), "max_rel_loss_dev": float(dev_loss.max()), "mean_rel_loss_dev": float(dev_loss.mean()),
                    "argmax_point": int(dev_loss.argmax()), "max_grad_dev_rel_to_largest_component": float(dev_grad.max()),
                    "bar": 1e-5, "meets_bar": bool(dev_loss.max() <= 1e-5),
                    "against": "fp64 instance of this library on the same lattice (oracle-checked at N=8192: tests/test_gpu_parity.py)"}

    if rank == 0 and args.config in ("trunc", "sparse"):
        ms_per_step = elapsed / args.steps * 1e3
        log(f"timed region done: {ms_per_step:.3f} ms/step")
        out = projected_report(args, eng, step, res, N, d, ntilde, n_kept, ms_per_step, units_per_step * args.steps / elapsed, world,
                               X, Xt, r, Bq, m_b, V_b, th1, lower, upper, grid, logA, lam0)
        print(json.dumps(out), flush=True)
        eng = None                                          # owned by the drop-in module's pool
    elif rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        log(f"timed region done: {ms_per_step:.2f} ms/step (host enqueue {eng.last_enqueue_ms():.2f} ms per unit)")
        units_per_s = units_per_step * args.steps / elapsed
        F = fits_flops(N, d)
        # ---- roofline of the dominant kernel, measured live on one more unit of this rank:
        #      HIP events around every launch of the GEMM family on the launching stream
        if args.config == "cells64":
            c0 = multi.partition(args.cells, world, rank)[0]
            rc, mc, Vc, thc = inputs[c0]
            probe = lambda: eng.fit_eval(thc, lower, upper, grid, X, rc, mc, Vc, logA, lam0, want_grad=want_grad, want_vectors=False)
        elif args.config == "thetagrid":
            probe = lambda: eng.fit_eval(points[0], lower, upper, grid, Xd, rd, md, Vd, logA, lam0, want_grad=want_grad, want_vectors=False,
                                         grad_precision=gprec)
        else:
            probe = step
        # (five profiled evaluations, their per-launch times averaged: one sample of the largest launch moves by +-4 %
        # with the clock state the chip happens to be in; counts and flops are the same every time)
        eng.set_profile(1)
        profs = []
        for _ in range(PROFILED_EVALUATIONS):
            probe()
            profs.append(eng.get_profile())
        prof = {k: (sum(p_[k] for p_ in profs) / len(profs) if isinstance(profs[0][k], float) else profs[0][k]) for k in profs[0]}
        phases = None
        if want_grad:
            eng.set_profile(2)
            probe()
            phases = eng.get_phases()
        executed_job = None
        if args.config == "thetagrid":
            # the timed lattice points reuse their context's V factor: their executed flops come from a probe that does too
            eng.set_profile(1)
            eng.fit_eval(points[0], lower, upper, grid, Xd, rd, md, Vd, logA, lam0, want_grad=want_grad, want_vectors=False,
                         grad_precision=gprec, reuse_V=True)
            pr = eng.get_profile()
            executed_job = pr["gemm_flops"] + pr["small_gemm_flops"] + pr["gram_flops"]
        eng.set_profile(0)
        unit_ms = phases["end"] if phases else ms_per_step
        gemm_tflops = prof["gemm_flops"] / (prof["gemm_ms"] * 1e-3) / 1e12 if prof["gemm_ms"] > 0 else 0.0
        # dominant kernel = the single largest launch: T = L^-1 L_V (both operands lower triangular, lower
        # output; algorithmic flops N^3/3): one launch per fit of
        # gemm_epi_kernel<R, false, true, 2> (the 128-tile body on the XCD-aware schedule, gemm_sched.hip, with the
        # tile-norm epilogue that leaves ||T||_F^2 behind; gemm_xcd_kernel<R, false, true> when GPFIT_FUSED_EPI
        # switches that epilogue off), so its
        # rocprofv3 kernel_stats row is this launch's average.
        npad = -(-N // 128) * 128
        nt = npad // 128
        dom_flops = float(npad) ** 3 / 3.0
        rname = "double" if dtype_name == "f64" else "float"   # the dominant launch (T) runs in fp32 in the mixed mode
        if dtype_name == "mixed":
            peak = FP32_MFMA_PEAK_TFLOPS
        # Which kernel T really is depends on its tile count (gemm.hip: launch_gemm): the XCD-aware data-parallel
        # schedule from 1536 tiles (with the tile-norm epilogue unless GPFIT_FUSED_EPI switches it off), the stream-K
        # schedule from 384, a small-tile launch below -- where no 128-tile launch exists and nothing is reported.
        t_tiles = nt * (nt + 1) // 2
        fused_norm = int(os.environ.get("GPFIT_FUSED_EPI", "7")) & 2
        if t_tiles >= 1536:
            dom_name = f"gemm_epi_kernel<{rname}, false, true, 2>" if fused_norm else f"gemm_xcd_kernel<{rname}, false, true>"
            dom_how = "XCD-aware macro-tile schedule"
        elif t_tiles >= 384:
            dom_name = f"gemm_streamk_kernel<{rname}, false, true, {2 if fused_norm else 0}>"
            dom_how = "stream-K schedule" + (" with the tile-norm epilogue" if fused_norm else "") + "; the live figure includes its fix-up kernel"
        else:
            dom_name, dom_how = None, None
        measured = prof["largest_gemm_ms"] > 0 and dom_name is not None
        dom_tflops = dom_flops / prof["largest_gemm_ms"] / 1e9 if measured else None
        executed = prof["gemm_flops"] + prof["small_gemm_flops"] + prof["gram_flops"]
        # the committed PMC passes (traffic, matrix-pipe utilisation, shader clock) describe the headline run only
        pmc_applies = measured and dtype_name == "f64" and args.config == "headline" and (N, d) == (8192, 256)
        util = profiled_mfma_util(dom_name) if pmc_applies else None
        traffic = profiled_traffic(dom_name, blocks=None) if pmc_applies else None
        notes = []
        if not measured:
            notes.append(f"T = L^-1 L_V has {t_tiles} 128-tiles at N={N}: no 128-tile launch of its own to time (small-tile instances), "
                         "achieved / frac are null; see gemm_family and unit_executed_frac")
        if measured and not pmc_applies:
            notes.append("traffic / mfma_util / clock are PMC passes of the headline run (N=8192 d=256 f64) and are not copied to other runs")
        if dtype_name == "mixed":
            notes.append("mixed precision: T and the other gradient products run on the fp32 MFMA (peak 157.3), the factorisations on the "
                         "fp64 MFMA (peak 78.6); unit_executed_frac is quoted against the fp32 peak and therefore understates the fp64 half")
        roofline = {
            "bound": "mfma",
            "kernel": None if dom_name is None else dom_name + " (T = L^-1 L_V, N^3/3 flops, 1 launch/fit, %s; %s)"
                      % (dom_how, "v_mfma_f64_16x16x4_f64" if dtype_name == "f64" else "v_mfma_f32_16x16x4_f32"),
            "note": " | ".join(notes) or None,
            "achieved": None if dom_tflops is None else round(dom_tflops, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": None if dom_tflops is None else round(dom_tflops / peak, 4),
            # HBM-side bytes per launch (fetch with the gfx950 correction + write) from the committed PMC passes; the
            # algorithmic operand bytes of the launch are 3 N^2 / 2 x 8 B (two triangular inputs, one triangular output)
            "traffic": None if traffic is None else round((traffic["fetch_corrected"] + traffic["write"]) * 1e9),
            "traffic_detail": traffic,
            "algorithmic_bytes_per_launch": 1.5 * float(npad) ** 2 * (8 if dtype_name == "f64" else 4),
            "mfma_util": util,
            "launches_per_fit": 1 if measured else 0, "avg_launch_ms": round(prof["largest_gemm_ms"], 4) if measured else None,
            "avg_over_profiled_evaluations": PROFILED_EVALUATIONS,
            "algorithmic_flops_per_launch": dom_flops,
            # `peak` is the nominal-clock figure (2.4 GHz).  The PMC pass shows the shader clock this launch
            # really ran at (the fp64 GEMMs sit near 2.07 GHz in steady state, profiles/r02_clock_probe.txt);
            # frac_at_measured_clock prices the same launch against the matrix peak at that clock.
            "clock": None if not (util and util.get("shader_clock_ghz")) else {
                "shader_clock_ghz": util["shader_clock_ghz"], "nominal_ghz": NOMINAL_GHZ,
                "peak_at_measured_clock": round(peak * util["shader_clock_ghz"] / NOMINAL_GHZ, 2),
                "frac_at_measured_clock": round(dom_tflops / (peak * util["shader_clock_ghz"] / NOMINAL_GHZ), 4),
                "source": util["source"]},
            "gemm_family": {"what": "all 128-tile GEMM/SYRK/TRSM/TRTRI launches (gemm_mfma_kernel<..,128> + gemm_streamk_kernel), executed flops",
                            "launches_per_fit": prof["gemm_launches"], "tflops": round(gemm_tflops, 2),
                            "frac": round(gemm_tflops / peak, 4),
                            "avg_launch_ms": round(prof["gemm_ms"] / max(1, prof["gemm_launches"]), 4)},
            # whole unit: flops the implementation EXECUTES (2.77 N^3 formulation, tile granularity) over the
            # unit's wall time -- the honest utilisation of the matrix peak by the unit of work
            "flops_executed_per_fit": executed,
            "unit_ms": round(unit_ms, 3),
            "unit_executed_tflops": round(executed / (unit_ms * 1e-3) / 1e12, 2),
            "unit_executed_frac": round(executed / (unit_ms * 1e-3) / 1e12 / peak, 4),
            # configurations of many units: the unit_* fields above describe ONE unit evaluated on its own (the profiled
            # probe); the timed job runs them in groups, and its utilisation is the same executed flops per unit times the
            # measured units per second (all GPUs)
            "job_executed_tflops": None if units_per_step <= 1 else round((executed_job or executed) * units_per_s / 1e12 / world, 2),
            "job_executed_frac": None if units_per_step <= 1 else round((executed_job or executed) * units_per_s / 1e12 / world / peak, 4),
            "job_flops_executed_per_unit": None if units_per_step <= 1 else (executed_job or executed),
            "phases_ms": phases,
            "gemm_ms_per_fit": round(prof["gemm_ms"], 3), "leaf_ms_per_fit": round(prof["leaf_ms"], 3),
            "leaf_launches_per_fit": prof["leaf_launches"],
            "small_tile_gemm": {"launches_per_fit": prof["small_gemm_launches"], "ms_per_fit": round(prof["small_gemm_ms"], 3),
                                "tflops": round(prof["small_gemm_flops"] / max(prof["small_gemm_ms"], 1e-9) / 1e9, 2)},
            "gram_ms_per_fit": round(prof["gram_ms"], 3),
            # SURVEY 8(d)'s algorithmic flop count F_fit = (14/3)N^3 + ... against the flops this formulation executes:
            # a saving of the FORMULATION (a speed-up factor), never a rate or a fraction of peak
            "unit_algorithmic_flops": F,
            "formulation_speedup_algorithmic_over_executed_flops": round(F / executed, 3) if executed > 0 else None,
        }
        metric = {"headline": f"GP fits/sec (kernel+chol+solve+grad loglik) at N={N} d={d}",
                  "n4096": f"GP fits/sec (kernel+chol+solve+grad loglik) at N={N} d={d}",
                  "cells64": "cells/sec, 64 independent cells x N=4096 d=128 sharded over the GPUs",
                  "thetagrid": "theta-points/sec, 512-point hyperparameter grid x N=8192 d=256 with gradients"}[args.config]
        workload = {"headline": f"N={N} d={d} single cell {dtype_name}, one M-step closure evaluation with 6 gradients"
                                + (" (BASELINE configs[2], headline)" if (N, d) == (8192, 256) else " (size override of the headline configuration)"),
                    "n4096": f"N={N} d={d} single cell {dtype_name}, one M-step closure evaluation with 6 gradients (BASELINE configs[1])",
                    "cells64": f"{args.cells} independent cells x N={N} d={d} {dtype_name}, cyclic shard over the ranks, X broadcast once, "
                               + (f"groups of {args.group} cells per call (lock-step factorisations, batched launches), {args.sets} set(s) of engines" if args.group > 0
                                  else f"{max(1, args.depth)} cells in flight per GPU") + " (BASELINE configs[3])",
                    "thetagrid": f"{args.grid_points} theta points x N={N} d={d} {dtype_name} with gradients, cyclic shard over the ranks, "
                                 f"X, r, m, V broadcast once, V factor reused across points, "
                                 + (f"groups of {args.group} points per call (lock-step factorisations, batched launches), {args.sets} set(s) of engines" if args.group > 0
                                    else f"{max(1, args.depth)} points in flight per GPU") + " (BASELINE configs[4])"}[args.config]
        if not want_grad:
            workload += " [forward only]"
        out = {
            "metric": metric + ("" if dtype_name == ("mixed" if args.config == "thetagrid" else "f64") else f" [{dtype_name} instance]"),
            "value": round(units_per_s, 4), "unit": f"{unit_name}/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None,
            "dtype": {"f64": "f64", "f32": "f32", "mixed": "f64 (kernel build, Cholesky, loss) + f32 (gradient products)"}[dtype_name],
            "data": "synthetic",
            "config": {"workload": workload, "N": N, "d": d, "units_per_step": units_per_step,
                       "parallelism": f"independent units over {world} GPU(s), one process per GPU, no data-path collective"
                                      + (f"; rank 0 holds {len(mine)} of {units_per_step} units: groups of {args.group} on "
                                         f"{args.sets} set(s) of contexts (asked for: {args.group_asked} x {args.sets_asked}, "
                                         "clamped to the local share)"
                                         if args.config in ("cells64", "thetagrid") and args.group > 0 else "")},
            "loss": res["loss"],
            "roofline": roofline,
        }
        if accuracy is not None:
            out["accuracy_vs_fp64"] = accuracy
        if world == 1 and args.config == "headline" and not args.no_in_flight:
            # Beside `value` (evaluations of ONE cell, each waiting for the previous one, as an L-BFGS closure does):
            # what the same GPU delivers when the evaluations are independent (several cells of this size, as in
            # configs[3]) and three are kept in flight on three contexts.  Informational, never `value`.
            out["independent_units_in_flight"] = in_flight_rate(eng, N, d, X, grid, dev, tdt, lower, upper, logA, lam0,
                                                                want_grad, gprec, local_rank)
        if world == 1 and not args.no_cpu_baseline and dtype_name == "f64" and args.config == "headline":
            cores, small, full = cpu_baseline(N, d, min(args.cpu_sample_n, N), args.cpu_quick)
            scale = (N / small["n"]) ** 3
            if full is not None:
                t_unit, how = full["t_ref"], (f"ONE real evaluation at N={N} d={d}: {full['t_ref']:.1f} s; plus 3 timed repeats (after a warm-up) at "
                                              f"N={small['n']}: {small['t_ref_all']} s (median {small['t_ref']:.2f} s, x{scale:.0f} by N^3 = {small['t_ref'] * scale:.1f} s)")
            else:
                t_unit, how = small["t_ref"] * scale, (f"3 timed repeats (after a warm-up) at N={small['n']} d={d}: {small['t_ref_all']} s, "
                                                      f"median {small['t_ref']:.2f} s scaled x{scale:.0f} (N^3) to N={N}")
            gpu_fits = units_per_s / world
            out["cpu_baseline"] = {
                "value": round(1.0 / t_unit, 6), "unit": "fits/s", "cores": cores, "kind": "port",
                "sample": "reference-formulation closure (oracle.mstep_closure_reference: materialised dK{6}, eigen-projection, "
                          "LU inverse, 13+13 gradient GEMMs; torch CPU fp64): " + how,
                "seconds_per_eval_at_sample_n": small["t_ref_all"],
                "seconds_per_eval_full_n": round(full["t_ref"], 2) if full else None,
                "cholesky_port_seconds_per_eval": {"n": (full or small)["n"], "s": round((full or small)["t_chol"], 3)},
                "gpu_vs_cpu": round(gpu_fits * t_unit, 1),
                "gpu_vs_cpu_cholesky_port": round(gpu_fits * (full or small)["t_chol"] * (1.0 if full else scale), 1),
            }
        print(json.dumps(out), flush=True)
    for e in extra_engines:
        e.close()
    if eng is not None and args.config not in ("trunc", "sparse"):
        eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
