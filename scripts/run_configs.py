"""Throughput of the non-headline BASELINE.json configurations, one JSON line each.

    python scripts/run_configs.py                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P scripts/run_configs.py          # N GPUs, one rank per GPU

 config1  N=4096, d=128 (16x8 pixel grid), single cell, fp64: fits/s
 config3  64 independent cells x N=4096 (d=128), cells sharded cyclically over the ranks,
          X broadcast once over RCCL: cells/s (whole job)
 config4  hyperparameter grid (8x8x8 lattice) x N=8192 (d=256) with gradients, fp32 instance of the
          library (gpfit_fit_eval_f32), theta points sharded over the ranks; V, m, r are shared by
          all points, so the V factor is reused (reuse_V); also run in fp64 for comparison: points/s.
"""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_processes_amd import multi, synthetic as syn
from gaussian_processes_amd.engine import GPFitEngine, fits_flops
from gaussian_processes_amd import utils as gp

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=64)
ap.add_argument("--lockstep", action="store_true", help="submit / collect the in-flight cells as groups (config3)")
ap.add_argument("--depth", type=int, default=4, help="independent cells kept in flight per GPU (config3)")
ap.add_argument("--grid-points", type=int, default=0, help="theta points to evaluate (default: 8 per rank)")
args = ap.parse_args()
world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); lrank = int(os.environ.get("LOCAL_RANK", "0"))
dist = None
if world > 1:
    import torch.distributed as dist
    torch.cuda.set_device(lrank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
dev = torch.device("cuda", lrank); torch.cuda.set_device(dev)
lower, upper = syn.limits()
logA, lam0 = syn.F_PARAMS["logA"], syn.F_PARAMS["lambda0"]


def tth(th):
    return {k: torch.tensor(v, dtype=torch.float64) for k, v in th.items()}


def kernel_half(X, grid, th0):
    C, mask = gp.localker(tth(th0), upper, lower, grid)
    Xm = X if bool(mask.all()) else X[:, mask].contiguous()
    return 0.5 * gp.acosker(tth(th0), Xm, Xm, C=C)


def sync_time():
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    return time.perf_counter()


def emit(**kw):
    if rank == 0:
        print(json.dumps(kw), flush=True)


def max_over_ranks(t):
    if dist is None:
        return t
    x = torch.tensor([t], dtype=torch.float64, device=dev)
    dist.all_reduce(x, op=dist.ReduceOp.MAX)
    return float(x)


# ---------------------------------------------------------------- config1: N=4096 d=128 single cell
N, d = 4096, 128
grid = (16, 8)
X = multi.broadcast_stimuli(torch.from_numpy(syn.stimuli(N, d)) if rank == 0 else None, (N, d), dev)
X = X.to(dev)
eng = GPFitEngine(N, d, device=lrank)
r_np, m_np = syn.cell_inputs(N, rank)
r, m = torch.from_numpy(r_np).to(dev), torch.from_numpy(m_np).to(dev)
V = kernel_half(X, grid, syn.theta0(rank))
th1 = syn.theta_eval(rank)
for _ in range(2):
    eng.fit_eval(th1, lower, upper, grid, X, r, m, V, logA, lam0, want_vectors=False)
t0 = sync_time(); K = 20
for _ in range(K):
    res = eng.fit_eval(th1, lower, upper, grid, X, r, m, V, logA, lam0, want_vectors=False)
el = max_over_ranks(sync_time() - t0)
emit(config="N=4096 d=128 single cell fp64", metric="fits/s", value=round(world * K / el, 3), n_gpus=world,
     ms_per_fit=round(el / K * 1e3, 3), unit_achieved_tflops=round(fits_flops(N, d) * K / el / 1e12, 2), loss=res["loss"])

# ---------------------------------------------------------------- config3: 64 cells x N=4096
cells = args.cells
mine = multi.partition(cells, world, rank)
inputs = {}
for c in mine:  # per-cell inputs built outside the timed region
    rc, mc = syn.cell_inputs(N, c)
    inputs[c] = (torch.from_numpy(rc).to(dev), torch.from_numpy(mc).to(dev), kernel_half(X, grid, syn.theta0(c)), syn.theta_eval(c))


def eval_cell(c):
    rc, mc, Vc, thc = inputs[c]
    o = eng.fit_eval(thc, lower, upper, grid, X, rc, mc, Vc, logA, lam0, want_vectors=False)
    return [o["loss"]] + [o["grad"][k] for k in syn.THETA_KEYS]


eval_cell(mine[0])
t0 = sync_time()
table = multi.run_sharded(cells, eval_cell, dev)
el = max_over_ranks(sync_time() - t0)
emit(config=f"{cells} independent cells x N=4096 d=128, cyclic shard, X broadcast once", metric="cells/s",
     value=round(cells / el, 3), n_gpus=world, seconds=round(el, 3), finite=bool(torch.isfinite(table).all()))

# same cells, two in flight on two contexts / streams (asynchronous entry point)
engs = [eng] + [GPFitEngine(N, d, device=lrank) for _ in range(args.depth - 1)]
streams = [torch.cuda.Stream(device=dev) for _ in engs]


def submit_cell(c, slot):
    rc, mc, Vc, thc = inputs[c]
    with torch.cuda.stream(streams[slot]):
        return engs[slot].fit_eval_async(thc, lower, upper, grid, X, rc, mc, Vc, logA, lam0, want_vectors=False)


def collect_cell(ticket, slot):
    o = engs[slot].fit_eval_finish(ticket)
    return [o["loss"]] + [o["grad"][k] for k in syn.THETA_KEYS]


for sl in range(1, args.depth):
    collect_cell(submit_cell(mine[0], sl), sl)
t0 = sync_time()
table2 = multi.run_sharded(cells, None, dev, submit_fn=submit_cell, collect_fn=collect_cell, depth=args.depth,
                           lockstep=args.lockstep)
el = max_over_ranks(sync_time() - t0)
emit(config=f"{cells} independent cells x N=4096 d=128, cyclic shard, X broadcast once, {args.depth} cells in flight per GPU{' (lock-step groups)' if args.lockstep else ''}",
     metric="cells/s", value=round(cells / el, 3), n_gpus=world, seconds=round(el, 3),
     identical_to_sequential=bool(torch.equal(table, table2)))
for e in engs[1:]:
    e.close()
del inputs, eng, engs, V
torch.cuda.empty_cache()

# ---------------------------------------------------------------- config4: theta grid x N=8192 (fp32 and fp64)
N, d = 8192, 256
grid = syn.grid_for(d)
X = multi.broadcast_stimuli(torch.from_numpy(syn.stimuli(N, d)) if rank == 0 else None, (N, d), dev).to(dev)
eng = GPFitEngine(N, d, device=lrank)
r_np, m_np = syn.cell_inputs(N, 0)
r, m = torch.from_numpy(r_np).to(dev), torch.from_numpy(m_np).to(dev)
V = kernel_half(X, grid, syn.theta0())
points = syn.theta_grid(8)
npts = args.grid_points or 8 * world
points = points[:npts]
mine = multi.partition(npts, world, rank)
ref_table = None
for dtype in (torch.float64, torch.float32):
    Xd, rd, md, Vd = (t.to(dtype) for t in (X, r, m, V))
    first = [True]

    def eval_point(u):
        o = eng.fit_eval(points[u], lower, upper, grid, Xd, rd, md, Vd, logA, lam0, want_vectors=False,
                         reuse_V=not first[0])
        first[0] = False
        return [o["loss"]] + [o["grad"][k] for k in syn.THETA_KEYS]

    eval_point(mine[0])
    t0 = sync_time()
    table = multi.run_sharded(npts, eval_point, dev)
    el = max_over_ranks(sync_time() - t0)
    extra = {}
    if ref_table is None:
        ref_table = table
    else:
        extra["max_rel_dev_of_loss_vs_fp64"] = float(((table[:, 0] - ref_table[:, 0]).abs() / ref_table[:, 0].abs()).max())
        extra["max_rel_dev_of_grad_vs_fp64"] = float((table[:, 1:] - ref_table[:, 1:]).abs().max() / ref_table[:, 1:].abs().max())
    emit(config=f"hyperparameter grid: {npts} of 512 theta points x N=8192 d=256 with gradients, "
                f"{'fp64' if dtype == torch.float64 else 'fp32'}, V factor reused",
         metric="theta-points/s", value=round(npts / el, 3), n_gpus=world, seconds=round(el, 3),
         finite=bool(torch.isfinite(table).all()), **extra)
if dist is not None:
    dist.destroy_process_group()
