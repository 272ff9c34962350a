"""Groups of independent units in one call (gpfit_fit_eval_batch) against unit-by-unit and pipelined evaluation.
    python scripts/scratch/dev_group.py [N] [d] [units] [group] [mode: f64|mixed] [reuse_V 0|1]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaussian_processes_amd import synthetic as syn, multi
from gaussian_processes_amd.engine import GPFitEngine, fit_eval_group
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
units = int(sys.argv[3]) if len(sys.argv) > 3 else 16
group = int(sys.argv[4]) if len(sys.argv) > 4 else 8
mode = sys.argv[5] if len(sys.argv) > 5 else "f64"
reuse = bool(int(sys.argv[6])) if len(sys.argv) > 6 else False
gprec = "f32" if mode == "mixed" else "native"
dev = torch.device("cuda:0")
grid = syn.grid_for(d); lower, upper = syn.limits()
X = torch.from_numpy(syn.stimuli(N, d)).to(dev)
logA, lam0 = syn.F_PARAMS["logA"], syn.F_PARAMS["lambda0"]
if reuse:   # theta grid: one cell, shared r, m, V
    rc, mc = syn.cell_inputs(N, 0)
    r0, m0, V0 = torch.from_numpy(rc).to(dev), torch.from_numpy(mc).to(dev), bench.build_V(X, grid, syn.theta0(), dev)
    pts = syn.theta_grid(8)
    inputs = [(r0, m0, V0, pts[(37 * u) % len(pts)]) for u in range(units)]
else:
    inputs = []
    for c in range(units):
        rc, mc = syn.cell_inputs(N, c)
        inputs.append((torch.from_numpy(rc).to(dev), torch.from_numpy(mc).to(dev), bench.build_V(X, grid, syn.theta0(c), dev), syn.theta_eval(c)))
engs = [GPFitEngine(N, d) for _ in range(group)]
key = lambda o: (float(o["loss"]).hex(),) + tuple(float(v).hex() for v in o["grad"].values())

def one_by_one():
    out = []
    for u, (r, m, V, th) in enumerate(inputs):
        out.append(engs[0].fit_eval(th, lower, upper, grid, X, r, m, V, logA, lam0, want_vectors=False, grad_precision=gprec,
                                    reuse_V=reuse and u > 0))
    return out

def grouped():
    out = []
    for g0 in range(0, units, group):
        sel = inputs[g0:g0 + group]
        out += fit_eval_group(engs, [t[3] for t in sel], lower, upper, grid, X, [t[0] for t in sel], [t[1] for t in sel],
                              [t[2] for t in sel], logA, lam0, grad_precision=gprec, reuse_V=reuse and g0 > 0 or (reuse and grouped.warm))
    grouped.warm = True
    return out
grouped.warm = False

def timed(fn, reps=2):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): res = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps, res

t1, a = timed(one_by_one)
if os.environ.get("DEV_GROUP_STREAM"):
    with torch.cuda.stream(torch.cuda.Stream(device=dev)):
        t2, b = timed(grouped)
else:
    t2, b = timed(grouped)
same = all(key(x) == key(y) for x, y in zip(a, b))
print(f"N={N} d={d} units={units} group={group} mode={mode} reuse_V={reuse} streams={os.environ.get('GPFIT_BATCH_STREAMS', 'default')}: "
      f"one by one {t1 / units * 1e3:.3f} ms/unit ({units / t1:.1f}/s); grouped {t2 / units * 1e3:.3f} ms/unit ({units / t2:.1f}/s); bit-identical {same}")
if not same:
    for i, (x, y) in enumerate(zip(a, b)):
        if key(x) != key(y): print("  unit", i, x["loss"], y["loss"], [x["grad"][k] - y["grad"][k] for k in x["grad"]]); break
