"""Grouped evaluation of independent units (gpfit_fit_eval_batch, through the C ABI) and the lock-step
factorisation behind it.  Run with `-m gpu` on an MI355X.

The claim under test is exactness, not a tolerance: a unit evaluated in a group, a unit evaluated alone, and a unit
evaluated with every launch on its own (the two free-running factorisation chains of rounds 1-2, no pointer
batches) produce the same bits, because every chain runs the same products in the same order and all
data-parallel GEMM instances sum k in ascending order per element.  The oracle comparison of the single path
(tests/test_gpu_parity.py) therefore carries over to the groups."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from gaussian_processes_amd import _lib, synthetic as syn
from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu
KEYS = syn.THETA_KEYS
LOWER, UPPER = syn.limits()
LOGA, LAM0 = syn.F_PARAMS["logA"], syn.F_PARAMS["lambda0"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the GPU"
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float64))


def cells(N, d, n_cells, dev, dtype=torch.float64):
    grid = syn.grid_for(d)
    X = T(syn.stimuli(N, d))
    out = []
    for c in range(n_cells):
        r_np, m_np = syn.cell_inputs(N, c)
        th0 = syn.theta0(c)
        C0, mask0 = orc.spatial_metric(th0, LOWER, UPPER, grid)
        V = 0.5 * orc.arccos_gram(th0, X[:, mask0], X[:, mask0], C0)
        out.append((T(r_np).to(dev).to(dtype), T(m_np).to(dev).to(dtype), V.to(dev).to(dtype), syn.theta_eval(c)))
    return grid, X.to(dev).to(dtype), out


def key(o):
    return (float(o["loss"]).hex(), float(o["loglik"]).hex(), float(o["KL"]).hex()) + tuple(float(o["grad"][k]).hex() for k in KEYS) \
        + (float(o["logdet_K"]).hex(), float(o["logdet_V"]).hex(), float(o["tr_KinvV"]).hex(), float(o["mKinvm"]).hex())


@pytest.fixture(scope="module")
def engines():
    from gaussian_processes_amd.engine import GPFitEngine
    made = {}

    def get(n, d, count):
        have = made.setdefault((n, d), [])
        while len(have) < count:
            have.append(GPFitEngine(n, d))
        return have[:count]
    yield get
    for lst in made.values():
        for e in lst:
            e.close()


@pytest.mark.parametrize("N,d,units", [(700, 64, 5), (1536, 128, 3), (256, 64, 16)])
def test_group_equals_unit_by_unit(dev, engines, N, d, units):
    """Units with different r, m, V, theta (ragged N, a rectangular pixel grid at d = 128, the largest group) in one
    call against the same units one by one on one engine: every output scalar bit for bit."""
    from gaussian_processes_amd.engine import fit_eval_group
    grid, X, inp = cells(N, d, units, dev)
    engs = engines(N, d, units)
    alone = [engs[0].fit_eval(th, LOWER, UPPER, grid, X, r, m, V, LOGA, LAM0, want_vectors=False) for r, m, V, th in inp]
    grouped = fit_eval_group(engs, [t[3] for t in inp], LOWER, UPPER, grid, X, [t[0] for t in inp], [t[1] for t in inp],
                             [t[2] for t in inp], LOGA, LAM0)
    assert [key(a) for a in alone] == [key(g) for g in grouped]
    assert len({a["loss"] for a in alone}) == units          # the units really are different problems
    # forward only
    fwd = fit_eval_group(engs, [t[3] for t in inp], LOWER, UPPER, grid, X, [t[0] for t in inp], [t[1] for t in inp],
                         [t[2] for t in inp], LOGA, LAM0, want_grad=False)
    assert [f["loss"] for f in fwd] == [a["loss"] for a in alone] and all(v == 0.0 for f in fwd for v in f["grad"].values())


def test_group_matches_oracle(dev, engines):
    """One group against the CPU oracle directly (tolerances of tests/test_gpu_parity.py)."""
    from gaussian_processes_amd.engine import fit_eval_group
    N, d, units = 300, 64, 4
    grid, X, inp = cells(N, d, units, dev)
    grouped = fit_eval_group(engines(N, d, units), [t[3] for t in inp], LOWER, UPPER, grid, X, [t[0] for t in inp],
                             [t[1] for t in inp], [t[2] for t in inp], LOGA, LAM0)
    for (r, m, V, th), g in zip(inp, grouped):
        loss, grad = orc.mstep_closure_cholesky(th, LOWER, UPPER, grid, X.cpu(), r.cpu(), m.cpu(), V.cpu(), LOGA, LAM0)
        assert abs(g["loss"] - loss) <= 1e-9 * abs(loss)
        ref = np.array([grad[k] for k in KEYS]); got = np.array([g["grad"][k] for k in KEYS])
        assert np.abs(ref - got).max() <= 1e-6 * np.abs(ref).max()


def test_group_mixed_precision_reuse_and_f32(dev, engines):
    """The theta-grid shape of a group: one cell's (r, m, V) shared by every unit, only theta differs; mixed
    precision with the V factor reused from the previous group, and the all-fp32 instance -- each against the
    same evaluations one by one."""
    from gaussian_processes_amd.engine import fit_eval_group
    N, d, units = 1024, 64, 4
    grid, X, inp = cells(N, d, 1, dev)
    r, m, V, _ = inp[0]
    pts = syn.theta_grid(8)
    thetas = [pts[(53 * u + 7) % len(pts)] for u in range(2 * units)]
    engs = engines(N, d, units)
    # mixed, second group reuses every context's V factor
    first = fit_eval_group(engs, thetas[:units], LOWER, UPPER, grid, X, r, m, V, LOGA, LAM0, grad_precision="f32")
    second = fit_eval_group(engs, thetas[units:], LOWER, UPPER, grid, X, r, m, V, LOGA, LAM0, grad_precision="f32", reuse_V=True)
    alone = [engs[0].fit_eval(th, LOWER, UPPER, grid, X, r, m, V, LOGA, LAM0, want_vectors=False, grad_precision="f32") for th in thetas]
    assert [key(a) for a in alone] == [key(g) for g in first + second]
    # all-fp32 instance
    X32, r32, m32, V32 = X.float(), r.float(), m.float(), V.float()
    g32 = fit_eval_group(engs, thetas[:units], LOWER, UPPER, grid, X32, r32, m32, V32, LOGA, LAM0)
    a32 = [engs[0].fit_eval(th, LOWER, UPPER, grid, X32, r32, m32, V32, LOGA, LAM0, want_vectors=False) for th in thetas[:units]]
    assert [key(a) for a in a32] == [key(g) for g in g32]
    assert max(abs(g["loss"] - a["loss"]) / abs(a["loss"]) for g, a in zip(g32, alone)) < 1e-3     # and it is the same problem


def test_group_with_a_unit_outside_the_limits_and_a_failed_cholesky(dev, engines):
    """utils.py:2020-2028 per unit: a theta outside the box gets the infinite loss / gradients and does not disturb
    the others; a unit whose V is not positive definite raises after every unit has been collected, and the
    engines stay usable."""
    from gaussian_processes_amd.engine import fit_eval_group
    N, d, units = 384, 64, 4
    grid, X, inp = cells(N, d, units, dev)
    engs = engines(N, d, units)
    thetas = [dict(t[3]) for t in inp]
    thetas[2]["eps_0x"] = float(UPPER["eps_0x"]) + 0.5
    res = fit_eval_group(engs, thetas, LOWER, UPPER, grid, X, [t[0] for t in inp], [t[1] for t in inp], [t[2] for t in inp], LOGA, LAM0)
    assert res[2]["loss"] == float("inf") and all(v == float("inf") for v in res[2]["grad"].values()) and not res[2]["in_bounds"]
    alone = [engs[0].fit_eval(t[3], LOWER, UPPER, grid, X, t[0], t[1], t[2], LOGA, LAM0, want_vectors=False) for t in inp]
    for u in (0, 1, 3):
        assert key(res[u]) == key(alone[u])
    Vbad = inp[1][2].clone()
    Vbad[5, 5] = -1.0
    Vs = [t[2] for t in inp]
    Vs[1] = Vbad
    with pytest.raises(_lib.GpfitError, match="Cholesky of V failed"):
        fit_eval_group(engs, [t[3] for t in inp], LOWER, UPPER, grid, X, [t[0] for t in inp], [t[1] for t in inp], Vs, LOGA, LAM0)
    again = fit_eval_group(engs, [t[3] for t in inp], LOWER, UPPER, grid, X, [t[0] for t in inp], [t[1] for t in inp],
                           [t[2] for t in inp], LOGA, LAM0)
    assert [key(a) for a in alone] == [key(g) for g in again]


def test_two_groups_in_flight_on_two_sets_of_engines(dev, engines):
    """fit_eval_group_begin / fit_eval_group_finish: a second group enqueued on other engines before the first is
    collected (each finish waits for its own group's completion event, not for the stream), through the pipelined
    driver of multi.evaluate_units_grouped -- same bits as one group at a time."""
    from gaussian_processes_amd import multi
    from gaussian_processes_amd.engine import fit_eval_group, fit_eval_group_begin, fit_eval_group_finish
    N, d, units, group = 640, 64, 10, 3
    grid, X, inp = cells(N, d, units, dev)
    engs = engines(N, d, 2 * group)

    def begin(us, slot):
        sel = [inp[u] for u in us]
        return fit_eval_group_begin(engs[slot * group:(slot + 1) * group], [t[3] for t in sel], LOWER, UPPER, grid, X,
                                    [t[0] for t in sel], [t[1] for t in sel], [t[2] for t in sel], LOGA, LAM0)

    def finish(handle, slot):
        return [[o["loss"]] + [o["grad"][k] for k in KEYS] for o in fit_eval_group_finish(handle)]

    piped = multi.evaluate_units_grouped(list(range(units)), None, dev, group, begin_fn=begin, finish_fn=finish, sets=2)
    plain = multi.evaluate_units_grouped(list(range(units)), lambda us: finish(begin(us, 0), 0), dev, group)
    assert torch.equal(piped, plain)
    alone = [engs[0].fit_eval(t[3], LOWER, UPPER, grid, X, t[0], t[1], t[2], LOGA, LAM0, want_vectors=False) for t in inp]
    assert [float(piped[u, 0]).hex() for u in range(units)] == [float(a["loss"]).hex() for a in alone]
    # a handle collected late, behind a later group on the same stream, still holds its own results
    h0 = begin([0, 1, 2], 0)
    h1 = begin([3, 4, 5], 1)
    r1, r0 = finish(h1, 1), finish(h0, 0)
    assert [r[0] for r in r0 + r1] == [float(piped[u, 0]) for u in range(6)]
    with pytest.raises(_lib.GpfitError, match="pending"):
        h = begin([0, 1, 2], 0)
        try:
            begin([3, 4, 5], 0)          # the same engines again before they were collected
        finally:
            finish(h, 0)


def test_group_argument_checks(dev, engines):
    from gaussian_processes_amd.engine import fit_eval_group, MAX_GROUP
    N, d = 256, 64
    grid, X, inp = cells(N, d, 2, dev)
    engs = engines(N, d, 2)
    r, m, V, th = inp[0]
    with pytest.raises(ValueError):
        fit_eval_group(engs, [th] * (MAX_GROUP + 1), LOWER, UPPER, grid, X, r, m, V, LOGA, LAM0)
    with pytest.raises(_lib.GpfitError, match="context of its own"):
        fit_eval_group([engs[0], engs[0]], [th, th], LOWER, UPPER, grid, X, r, m, V, LOGA, LAM0)
    with pytest.raises(ValueError, match="same shape"):
        fit_eval_group(engs, [th, th], LOWER, UPPER, grid, [X, X[:128]], r, m, V, LOGA, LAM0)


_CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
import numpy as np
from gaussian_processes_amd import synthetic as syn
from gaussian_processes_amd.engine import GPFitEngine
from oracle import gp_oracle as orc
N, d = 1408, 64
dev = torch.device("cuda:0")
grid = syn.grid_for(d); lower, upper = syn.limits()
X = torch.from_numpy(syn.stimuli(N, d))
r_np, m_np = syn.cell_inputs(N, 2)
C0, mask0 = orc.spatial_metric(syn.theta0(2), lower, upper, grid)
V = 0.5 * orc.arccos_gram(syn.theta0(2), X[:, mask0], X[:, mask0], C0)
eng = GPFitEngine(N, d)
o = eng.fit_eval(syn.theta_eval(2), lower, upper, grid, X.to(dev), torch.from_numpy(r_np).to(dev), torch.from_numpy(m_np).to(dev),
                 V.to(dev), syn.F_PARAMS["logA"], syn.F_PARAMS["lambda0"], want_vectors=False)
print("HEX", float(o["loss"]).hex(), " ".join(float(v).hex() for v in o["grad"].values()), float(o["logdet_V"]).hex(), float(o["tr_KinvV"]).hex())
"""


def test_lockstep_and_free_running_schedules_give_the_same_bits():
    """The single unit under the schedules of the factorisations -- lock step with shared and paired launches
    (default), without the paired launches (GPFIT_NO_PAIR: the update of a node and the first product of its inverse
    merge as two launches), with every product launched on its own (GPFIT_NO_BATCH), the two free-running chains of
    rounds 1-2 (GPFIT_LOCKSTEP=0, with and without its own paired launches) -- one process each (the switches are read
    once per process)."""
    outs = []
    for extra in ({}, {"GPFIT_NO_PAIR": "1"}, {"GPFIT_NO_BATCH": "1"}, {"GPFIT_LOCKSTEP": "0"}, {"GPFIT_LOCKSTEP": "0", "GPFIT_NO_PAIR": "1"}):
        env = dict(os.environ, **extra)
        p = subprocess.run([sys.executable, "-c", _CHILD % ROOT], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append([l for l in p.stdout.splitlines() if l.startswith("HEX")][0])
    assert all(o == outs[0] for o in outs), outs


_CHILD_SK = r"""
import sys, torch
sys.path.insert(0, %r)
from gaussian_processes_amd import synthetic as syn
from gaussian_processes_amd.engine import GPFitEngine
import bench
N, d = 3712, 64                       # 29 x 30 / 2 = 435 lower 128-tiles: T takes the stream-K schedule
dev = torch.device("cuda:0")
grid = syn.grid_for(d); lower, upper = syn.limits()
X = torch.from_numpy(syn.stimuli(N, d)).to(dev)
r_np, m_np = syn.cell_inputs(N, 1)
V = bench.build_V(X, grid, syn.theta0(1), dev)
eng = GPFitEngine(N, d)
o = eng.fit_eval(syn.theta_eval(1), lower, upper, grid, X, torch.from_numpy(r_np).to(dev), torch.from_numpy(m_np).to(dev), V,
                 syn.F_PARAMS["logA"], syn.F_PARAMS["lambda0"], want_vectors=False)
print("VAL", repr(float(o["tr_KinvV"])), repr(float(o["loss"])), " ".join(repr(float(v)) for v in o["grad"].values()))
"""


def test_tile_norms_on_the_stream_k_schedule_match_the_separate_pass():
    """tr(K~^-1 V) = ||T||_F^2 at a size where T = L^-1 L_V takes the stream-K schedule: the tile norms left behind
    by the launch and its fix-up kernel (33 table entries per tile, gemm_sumsq_entries) against the separate pass over
    T (GPFIT_FUSED_EPI=0) -- two processes, the switch is read once.  Same T, two summation orders of N^2 / 2 squares."""
    vals = []
    for extra in ({}, {"GPFIT_FUSED_EPI": "0"}):
        p = subprocess.run([sys.executable, "-c", _CHILD_SK % ROOT], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        vals.append([float(x) for x in [l for l in p.stdout.splitlines() if l.startswith("VAL")][0].split()[1:]])
    fused, separate = np.array(vals[0]), np.array(vals[1])
    assert abs(fused[0] - separate[0]) <= 1e-13 * abs(separate[0]), (fused[0], separate[0])      # the trace term itself
    assert abs(fused[1] - separate[1]) <= 1e-13 * abs(separate[1])                                # the loss it enters
    assert np.abs(fused[2:] - separate[2:]).max() <= 1e-11 * np.abs(separate[2:]).max()           # gradients: Q's passes differ too
