"""fp32 instance of the fused unit of work (hyperparameter-grid configuration, BASELINE
configs[4]) against the fp64 path on the same inputs.  The reference has no fp32 mode
(utils.py:31-33), so the fp64 result of this library -- oracle-checked in tests/test_gpu_parity.py -- is the
yardstick here (a comparison of two instances of the same library, not a pin to the reference).

Stated tolerances: loss 1e-5 relative (the north star's bar), its parts loglik / KL 2e-5 relative,
gradients 2e-3 of the largest component, posterior vectors 1e-4 relative.  cond(K~) ~ 5e5 at
N=8192, i.e. fp32 loses ~1e-2 in the weakest eigen-directions but the log-determinant, traces
and quadratic forms are dominated by the well-conditioned ones."""
import numpy as np
import pytest
import torch

from conftest import relerr
from gaussian_processes_amd import synthetic as syn
from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu
KEYS = syn.THETA_KEYS
LOWER, UPPER = syn.limits()
LOGA, LAM0 = syn.F_PARAMS["logA"], syn.F_PARAMS["lambda0"]


def case(N, d, dev):
    from gaussian_processes_amd import utils as gp
    grid = syn.grid_for(d)
    X = torch.from_numpy(syn.stimuli(N, d)).to(dev)
    r_np, m_np = syn.cell_inputs(N)
    r, m = torch.from_numpy(r_np).to(dev), torch.from_numpy(m_np).to(dev)
    t0 = {k: torch.tensor(v, dtype=torch.float64) for k, v in syn.theta0().items()}
    C, mask = gp.localker(t0, UPPER, LOWER, grid)
    V = 0.5 * gp.acosker(t0, X, X, C=C)
    return grid, X, r, m, V


@pytest.mark.parametrize("N,d", [(200, 16), (512, 64), (2048, 256), (4096, 128), (8192, 256)])
def test_fp32_unit_of_work_tracks_fp64(N, d):
    from gaussian_processes_amd.engine import GPFitEngine
    dev = torch.device("cuda:0")
    grid, X, r, m, V = case(N, d, dev)
    th1 = syn.theta_eval()
    eng = GPFitEngine(N, d)
    a = eng.fit_eval(th1, LOWER, UPPER, grid, X, r, m, V, LOGA, LAM0)
    f32 = lambda t: t.to(torch.float32)
    b = eng.fit_eval(th1, LOWER, UPPER, grid, f32(X), f32(r), f32(m), f32(V), LOGA, LAM0)
    eng.close()
    assert b["lam_m"].dtype == torch.float32
    assert abs(b["loss"] - a["loss"]) <= 1e-5 * abs(a["loss"]), (a["loss"], b["loss"])   # the north star's bar
    for key in ("loglik", "KL"):
        assert abs(b[key] - a[key]) <= 2e-5 * abs(a[key]), (key, a[key], b[key])
    ga = np.array([a["grad"][k] for k in KEYS]); gb = np.array([b["grad"][k] for k in KEYS])
    assert np.abs(ga - gb).max() <= 2e-3 * np.abs(ga).max(), (ga, gb)
    assert relerr(b["lam_var"].double().cpu().numpy(), a["lam_var"].cpu().numpy()) < 1e-4
    assert relerr(b["f"].double().cpu().numpy(), a["f"].cpu().numpy()) < 1e-4
    print(f"N={N} d={d}: loss rel {abs(b['loss']-a['loss'])/abs(a['loss']):.2e} grad rel {np.abs(ga-gb).max()/np.abs(ga).max():.2e}")


def stratified_lattice_points():
    """64 of the 512 lattice points of BASELINE configs[4] (8 x 8 x 8 over -2log2beta, -log2rho2, Amp,
    +-0.35 around theta0): the 8 corners plus 56 points that cover every level of every axis
    (a 4 x 4 x 4 sub-lattice on the even levels shifted cyclically, minus duplicates of corners)."""
    pts = syn.theta_grid(8)
    idx = lambda a, b, c: (a * 8 + b) * 8 + c
    chosen = [idx(a, b, c) for a in (0, 7) for b in (0, 7) for c in (0, 7)]
    for a in range(8):
        for b in range(8):
            c = (3 * a + 5 * b + 1) % 8           # a Latin-square style spread: every level of c for every a and b
            chosen.append(idx(a, b, c))
    chosen = list(dict.fromkeys(chosen))[:64]
    return [pts[i] for i in chosen], chosen


def test_theta_grid_meets_the_north_star_tolerance():
    """BASELINE configs[4] at its stated bar (north star: 'log-lik match <= 1e-5 rel') on ALL 512 points of the
    lattice, N = 8192, d = 256, as the grid driver of bench.py runs it: groups of eight points per call, every
    context reusing its own copy of the V factor.

    Yardstick: the fp64 instance of this library on the same lattice (the reference has no reduced-precision mode,
    utils.py:31-33; the fp64 instance is oracle-checked at this size in tests/test_gpu_parity.py) -- a yardstick,
    not a second pin.  The all-fp32 instance misses the bar on part of the lattice (2.8e-5 at corner 448; measured
    with scripts/scratch/dev_fp32_err.py: rounding K~, V, m to fp32 moves the loss by 1e-8 -- it is the fp32 ARITHMETIC of
    the factorisations, log|K~| off by +0.5 .. +1.7, that does it), so the configuration runs in the mixed mode:
    kernel build, both Cholesky factorisations, both log-determinants, the likelihood and m^T K~^-1 m in fp64; the
    N^3-heavy products T, Q, W and the pull-back in fp32 -- including T's norm, i.e. the trace term tr(K~^-1 V) of
    the loss is fp32-derived (its rounding errors average out over N^2 / 2 squares: measured 2e-9 on the loss).
    Asserted: mixed mode loss <= 1e-5 on every lattice point (the measured maximum is printed and kept two orders
    below the bar), gradients <= 1e-3 of the largest component (the north star states no gradient tolerance);
    all-fp32 loss <= 5e-5 on the 64 stratified points, a regression bound and not a claim of the bar."""
    from gaussian_processes_amd.engine import GPFitEngine, fit_eval_group
    dev = torch.device("cuda:0")
    N, d, group = 8192, 256, 8
    grid, X, r, m, V = case(N, d, dev)
    lattice = syn.theta_grid(8)
    assert len(lattice) == 512
    engs = [GPFitEngine(N, d) for _ in range(group)]

    def sweep(points, Xs, rs, ms, Vs, **kw):
        rows = []
        for g0 in range(0, len(points), group):
            res = fit_eval_group(engs, points[g0:g0 + group], LOWER, UPPER, grid, Xs, rs, ms, Vs, LOGA, LAM0, reuse_V=g0 > 0, **kw)
            rows += [[o["loss"]] + [o["grad"][k] for k in KEYS] for o in res]
        return np.array(rows)

    ref = sweep(lattice, X, r, m, V)
    mixed = sweep(lattice, X, r, m, V, grad_precision="f32")
    points, chosen = stratified_lattice_points()
    assert len(points) == 64 and {0, 7, 56, 63, 448, 455, 504, 511} <= set(chosen)
    all32 = sweep(points, *(t.to(torch.float32) for t in (X, r, m, V)))
    for e in engs:
        e.close()
    assert np.isfinite(ref).all() and len(np.unique(ref[:, 0])) == 512
    for name, got, base, loss_tol, ids in (("mixed", mixed, ref, 1e-5, list(range(512))), ("all-fp32", all32, ref[chosen], 5e-5, chosen)):
        loss_dev = np.abs(got[:, 0] - base[:, 0]) / np.abs(base[:, 0])
        grad_dev = np.abs(got[:, 1:] - base[:, 1:]).max(1) / np.abs(base[:, 1:]).max(1)
        print(f"{name} vs fp64 over {len(ids)} lattice points: loss max {loss_dev.max():.2e} (point {ids[int(loss_dev.argmax())]}), "
              f"median {np.median(loss_dev):.2e}; grad max {grad_dev.max():.2e}")
        assert loss_dev.max() <= loss_tol, (name, loss_dev.max(), ids[int(loss_dev.argmax())])
        assert grad_dev.max() <= 1e-3, (name, grad_dev.max())
    # drift guard: the mixed mode sits orders of magnitude below the bar; a regression towards it should be seen
    mixed_dev = np.abs(mixed[:, 0] - ref[:, 0]) / np.abs(ref[:, 0])
    assert mixed_dev.max() <= 1e-7, f"mixed-mode loss deviation grew to {mixed_dev.max():.2e} (was 2e-9)"


def test_mixed_precision_small_and_ragged():
    """The mixed mode on sizes with padding (N not a multiple of 128) and a rectangular pixel grid:
    the loss and its parts are the fp64 ones bit for bit (same kernels), gradients within 1e-3."""
    from gaussian_processes_amd.engine import GPFitEngine
    dev = torch.device("cuda:0")
    for N, d in ((200, 16), (1000, 128), (2048, 256)):
        grid, X, r, m, V = case(N, d, dev)
        eng = GPFitEngine(N, d)
        th1 = syn.theta_eval()
        a = eng.fit_eval(th1, LOWER, UPPER, grid, X, r, m, V, LOGA, LAM0)
        b = eng.fit_eval(th1, LOWER, UPPER, grid, X, r, m, V, LOGA, LAM0, grad_precision="f32")
        c2 = eng.fit_eval(th1, LOWER, UPPER, grid, X, r, m, V, LOGA, LAM0, grad_precision="f32", reuse_V=True)
        eng.close()
        assert a["loglik"] == b["loglik"] and a["logdet_K"] == b["logdet_K"] and a["logdet_V"] == b["logdet_V"]
        assert abs(a["loss"] - b["loss"]) <= 1e-7 * abs(a["loss"])          # tr(K~^-1 V) comes from the fp32 T
        ga = np.array([a["grad"][k] for k in KEYS])
        for o in (b, c2):
            gb = np.array([o["grad"][k] for k in KEYS])
            assert np.abs(ga - gb).max() <= 1e-3 * np.abs(ga).max(), (N, ga, gb)
        assert b["loss"] == c2["loss"]


def test_theta_points_in_flight_match_one_at_a_time():
    """The theta-grid driver of bench.py keeps three points in flight on three contexts, each reusing its own
    copy of the V factor (asynchronous entry point + reuse flag + mixed precision together): bit-identical to
    the one-at-a-time sweep on one context."""
    from gaussian_processes_amd import multi
    from gaussian_processes_amd.engine import GPFitEngine
    dev = torch.device("cuda:0")
    N, d, depth = 1000, 64, 3
    grid, X, r, m, V = case(N, d, dev)
    points = syn.theta_grid(8)[::73][:7]
    engs = [GPFitEngine(N, d) for _ in range(depth)]
    streams = [torch.cuda.Stream() for _ in engs]
    torch.cuda.synchronize()
    row = lambda o: [o["loss"]] + [o["grad"][k] for k in KEYS]
    for prec in ("native", "f32"):
        fresh = [True] * depth
        first = [True]

        def one(u):
            o = engs[0].fit_eval(points[u], LOWER, UPPER, grid, X, r, m, V, LOGA, LAM0, want_vectors=False,
                                 reuse_V=not first[0], grad_precision=prec)
            first[0] = False
            return row(o)

        def submit(u, slot):
            with torch.cuda.stream(streams[slot]):
                t = engs[slot].fit_eval_async(points[u], LOWER, UPPER, grid, X, r, m, V, LOGA, LAM0, want_vectors=False,
                                              reuse_V=not fresh[slot], grad_precision=prec)
            fresh[slot] = False
            return t

        seq = multi.run_sharded(len(points), one, dev)
        pipe = multi.run_sharded(len(points), None, dev, submit_fn=submit,
                                 collect_fn=lambda t, slot: row(engs[slot].fit_eval_finish(t)), depth=depth)
        assert torch.isfinite(seq).all() and torch.equal(seq, pipe), prec
    for e in engs:
        e.close()


def test_fp32_rejects_mixed_dtypes():
    from gaussian_processes_amd.engine import GPFitEngine
    dev = torch.device("cuda:0")
    grid, X, r, m, V = case(128, 16, dev)
    eng = GPFitEngine(128, 16)
    with pytest.raises(TypeError):
        eng.fit_eval(syn.theta_eval(), LOWER, UPPER, grid, X.float(), r, m, V, LOGA, LAM0)
    eng.close()
