"""Fit-model persistence with the reference's on-disk layout (``Spatial_GP_repo/utils.py:46-109``,
``312-324``): a directory holding ``model`` (pickle of the ``fit_model`` dict ``varGP`` returns, with a
``description`` entry added) and ``metadata`` (the description as text).  The notebooks call these right
after a fit (``one_cell_fit.ipynb``), so a drop-in ``utils`` needs them; nothing here touches the GPU
path -- tensors are moved to host memory before pickling so a model saved on one device loads anywhere.
"""
from __future__ import annotations

import math
import os
import pickle

import torch

_THETA_ROWS = ("sigma_0", "eps_0x", "eps_0y", "Amp", "-2log2beta", "-log2rho2")


def _first_last(track, key):
    seq = track[key]
    return float(seq[0]), float(seq[-1])


def describe(model) -> str:
    """Human-readable summary of a fit (what ``save_model`` stores under ``description``): the fit
    settings, start -> end of every hyperparameter (plus beta / rho of the paper's parametrisation,
    utils.py:726-734) and of the link-function parameters."""
    fp = model["fit_parameters"]
    tt = model["values_track"]["theta_track"]
    ft = model["values_track"]["f_par_track"]
    lines = ["Model Description:"]
    for label, key, fmt in (("Cell ID", "cellid", ">8"), ("ntilde", "ntilde", ">8"), ("maxiter", "maxiter", ">8"),
                            ("nMstep", "nMstep", ">8"), ("nEstep", "nEstep", ">8"),
                            ("MIN_TOLERANCE", "min_tolerance", ">8.12f"), ("EIGVAL_TOL", "eigval_tol", ">8.4f")):
        lines.append(f"{label + ':':<15}{format(fp[key], fmt)}")
    lines += ["", "Hyperparameters results:", "Start                 ->   End"]
    for key in _THETA_ROWS:
        a, b = _first_last(tt, key)
        lines.append(f"{key + ':':<13}{a:>8.4f} -> {b:>8.4f}")
    a, b = _first_last(tt, "-2log2beta")
    lines.append(f"{'beta:':<13}{0.5 * math.exp(-0.5 * a):>8.4f} -> {0.5 * math.exp(-0.5 * b):>8.4f}")
    a, b = _first_last(tt, "-log2rho2")
    lines.append(f"{'rho:':<13}{math.exp(-0.5 * a) / math.sqrt(2):>8.4f} -> {math.exp(-0.5 * b) / math.sqrt(2):>8.4f}")
    lines += ["", "Link function results [f_params]:"]
    a, b = _first_last(ft, "logA")
    lines.append(f"{'logA:':<13}{a:>8.4f} -> {b:>8.4f}")
    lines.append(f"{'A:':<13}{math.exp(a):>8.4f} -> {math.exp(b):>8.4f}")
    if "lambda0" in ft:
        a, b = _first_last(ft, "lambda0")
        lines.append(f"{'lambda0:':<13}{a:>8.4f} -> {b:>8.4f}")
    return "\n".join(lines) + "\n"


def _to_host(obj):
    if torch.is_tensor(obj):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return {k: _to_host(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_host(v) for v in obj)
    return obj


def save_model(model, directory, additional_description=None):
    """utils.py:46: refuses an existing directory (ValueError), adds ``model['description']``, writes
    ``<directory>/model`` and ``<directory>/metadata``."""
    if os.path.exists(directory):
        raise ValueError(f"Directory {directory} already exists")
    description = describe(model)
    if additional_description is not None:
        description += f"\n\n{additional_description}"
    os.makedirs(directory)
    model["description"] = description
    with open(os.path.join(directory, "model"), "wb") as f:
        pickle.dump(_to_host(model), f)
    with open(os.path.join(directory, "metadata"), "w") as f:
        f.write(description)


def load_model(directory, map_location=None):
    """utils.py:312: the dict ``save_model`` wrote.  ``map_location`` (an addition) moves every tensor to
    that device, e.g. ``'cuda:0'`` before handing the dict to ``test(**model)``."""
    with open(os.path.join(directory, "model"), "rb") as f:
        model = pickle.load(f)
    if map_location is not None:
        dev = torch.device(map_location)

        def move(obj):
            if torch.is_tensor(obj):
                return obj.to(dev)
            if isinstance(obj, dict):
                return {k: move(v) for k, v in obj.items()}
            if isinstance(obj, (list, tuple)):
                return type(obj)(move(v) for v in obj)
            return obj
        model = move(model)
    return model
