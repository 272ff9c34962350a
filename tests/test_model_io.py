"""save_model / load_model keep the reference's directory layout (utils.py:46-109, 312-324): 'model'
(pickle of the fit_model dict plus 'description') and 'metadata' (text); an existing directory is refused."""
import os

import pytest
import torch

from gaussian_processes_amd import model_io


def toy_model():
    keys = ("sigma_0", "eps_0x", "eps_0y", "-2log2beta", "-log2rho2", "Amp")
    tt = {k: torch.linspace(0.1 * (i + 1), 0.2 * (i + 1), 3, dtype=torch.float64) for i, k in enumerate(keys)}
    return {
        "fit_parameters": {"cellid": 8, "ntilde": 32, "maxiter": 3, "nMstep": 4, "nEstep": 5,
                           "min_tolerance": 1e-11, "eigval_tol": 1e-4},
        "values_track": {"theta_track": tt,
                         "f_par_track": {"logA": torch.tensor([-3.0, -2.9, -2.8]), "lambda0": torch.tensor([-0.3, -0.2, -0.1])}},
        "m_b": torch.arange(4, dtype=torch.float64), "V_b": torch.eye(4, dtype=torch.float64),
        "hyperparams_tuple": ({k: v[-1] for k, v in tt.items()}, None, None),
    }


def test_round_trip_layout_and_refusal(tmp_path):
    model = toy_model()
    target = os.path.join(tmp_path, "fit_cell8")
    model_io.save_model(model, target, additional_description="synthetic")
    assert sorted(os.listdir(target)) == ["metadata", "model"]
    text = open(os.path.join(target, "metadata")).read()
    assert "Cell ID:" in text and "synthetic" in text and "0.1000 ->   0.2000" in text       # sigma_0 start -> end
    assert "beta:" in text and "rho:" in text and "lambda0:" in text
    back = model_io.load_model(target, map_location="cpu")
    assert back["description"] == text == model["description"]
    assert torch.equal(back["V_b"], model["V_b"]) and torch.equal(back["m_b"], model["m_b"])
    assert torch.equal(back["values_track"]["theta_track"]["Amp"], model["values_track"]["theta_track"]["Amp"])
    assert isinstance(back["hyperparams_tuple"], tuple)
    with pytest.raises(ValueError, match="already exists"):
        model_io.save_model(model, target)


def test_out_of_scope_names_say_why():
    """Names of the reference's utils.py that this module leaves out raise an AttributeError that says so
    (a notebook cell calling the plotting helper gets an explanation, not a bare failure)."""
    pytest.importorskip("torch")
    from gaussian_processes_amd import utils as gp
    with pytest.raises(AttributeError, match="plotting helper"):
        gp.plot_loss_and_theta_notebook
    with pytest.raises(AttributeError, match="no attribute"):
        gp.definitely_not_a_name
    assert callable(gp.save_model) and callable(gp.load_model) and callable(gp.nd_utility)
