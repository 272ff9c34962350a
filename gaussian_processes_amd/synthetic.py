"""Deterministic synthetic workloads for the GP fit path.

The reference ships no data (its lab pickle is absent, ``data.py:6``); every
test, fixture and benchmark uses these seeded inputs instead.  numpy only, so the
very same arrays can be regenerated in the build container (where the reference
is importable) and on the GPU box (where it is not).
"""
from __future__ import annotations

import math

import numpy as np

THETA_KEYS = ("sigma_0", "eps_0x", "eps_0y", "-2log2beta", "-log2rho2", "Amp")


def grid_for(d: int):
    """Pixel grid (n_rows, n_cols) with n_rows*n_cols == d: square when d is a
    perfect square (the only case the reference supports, utils.py:876), otherwise
    the most balanced 2:1 rectangle (d=128 -> 16x8)."""
    s = int(round(math.sqrt(d)))
    if s * s == d:
        return (s, s)
    r = int(round(math.sqrt(2 * d)))
    if r * (d // r) == d:
        return (r, d // r)
    raise ValueError(f"no pixel grid for d={d}")


def theta0(cell: int = 0):
    """Base hyperparameters of SURVEY 8(d); ``cell`` moves the receptive-field centre."""
    th = {
        "sigma_0": 1.0,
        "eps_0x": 0.05,
        "eps_0y": -0.03,
        "-2log2beta": -2.0 * math.log(2 * 0.6),
        "-log2rho2": -math.log(2 * 0.3 ** 2),
        "Amp": 1.0,
    }
    if cell:
        th["eps_0x"] = -0.2 + 0.05 * (cell % 8)
        th["eps_0y"] = -0.2 + 0.05 * (cell // 8)
    return th


def theta_eval(cell: int = 0):
    """Evaluation point: theta0 with -log2rho2 += 0.05 and Amp *= 1.02."""
    th = theta0(cell)
    th["-log2rho2"] += 0.05
    th["Amp"] *= 1.02
    return th


def theta_grid(n_side: int = 8, span: float = 0.35):
    """n_side^3 lattice over (-2log2beta, -log2rho2, Amp) +-span around theta0 (config 5)."""
    base = theta0()
    offs = np.linspace(-span, span, n_side)
    out = []
    for a in offs:
        for b in offs:
            for c in offs:
                th = dict(base)
                th["-2log2beta"] += float(a)
                th["-log2rho2"] += float(b)
                th["Amp"] += float(c)
                out.append(th)
    return out


def limits():
    """Hyperparameter boxes of generate_theta (utils.py:854-855)."""
    inf = float("inf")
    lower = {"sigma_0": 0.0, "eps_0x": -1.0, "eps_0y": -1.0, "-2log2beta": -inf, "-log2rho2": -inf, "Amp": 0.0}
    upper = {"sigma_0": inf, "eps_0x": 1.0, "eps_0y": 1.0, "-2log2beta": inf, "-log2rho2": inf, "Amp": inf}
    return lower, upper


def stimuli(N: int, d: int, seed: int = 0):
    """Shared stimulus matrix X[N,d] ~ N(0,1), fp64."""
    return np.random.default_rng(seed).standard_normal((N, d))


def cell_inputs(N: int, cell: int = 0):
    """Per-cell responses r ~ Poisson(0.7) and variational mean m = 0.1*randn."""
    rng = np.random.default_rng(1000 + cell)
    r = rng.poisson(0.7, N).astype(np.float64)
    m = 0.1 * rng.standard_normal(N)
    return r, m


F_PARAMS = {"logA": math.log(0.05), "lambda0": -0.3}
