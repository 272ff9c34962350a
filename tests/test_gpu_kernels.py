"""GPU parity of the materialising entry points (localker, acosker and their derivatives,
through the drop-in module -> C ABI) against the golden vectors of the real reference."""
import numpy as np
import pytest
import torch

from conftest import load_golden, relerr
from gaussian_processes_amd import synthetic as syn

pytestmark = pytest.mark.gpu
KEYS = syn.THETA_KEYS
LOWER, UPPER = syn.limits()
DCK = ("Amp", "-2log2beta", "-log2rho2", "eps_0x", "eps_0y")


def tth(vec):
    return {k: torch.tensor(float(v), dtype=torch.float64) for k, v in zip(KEYS, vec)}


def T(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float64)).cuda()


def test_localker_matches_golden():
    from gaussian_processes_amd import utils as gp
    g = load_golden("g1_localker.npz")
    for i in range(int(g["n_cases"])):
        C, mask, dC = gp.localker(tth(g[f"c{i}_theta"]), UPPER, LOWER, int(g[f"c{i}_n_px"]), grad=True)
        assert np.array_equal(mask.cpu().numpy(), g[f"c{i}_mask"])
        assert relerr(C.cpu().numpy(), g[f"c{i}_C"]) < 1e-13
        for k in DCK:
            assert relerr(dC[k].cpu().numpy(), g[f"c{i}_dC_{k}"]) < 1e-13, k
        C2, mask2 = gp.localker(tth(g[f"c{i}_theta"]), UPPER, LOWER, int(g[f"c{i}_n_px"]), grad=False)
        assert torch.equal(C2, C)


def test_localker_raises_outside_limits():
    from gaussian_processes_amd import utils as gp
    th = tth([1.0, 1.2, 0.0, 0.0, 0.0, 1.0])
    with pytest.raises(ValueError, match="eps_0x"):
        gp.localker(th, UPPER, LOWER, 8)


def test_acosker_matches_golden():
    from gaussian_processes_amd import utils as gp
    g = load_golden("g2_acosker.npz")
    th = tth(g["theta"])
    C, mask, dC = gp.localker(th, UPPER, LOWER, int(g["n_px"]), grad=True)
    X, X2, X1 = T(g["X"])[:, mask], T(g["X2"])[:, mask], T(g["X1row"])[:, mask]
    # square, with derivatives
    K, dK = gp.acosker(th, X, X, C=C, dC=dC, diag=False)
    assert relerr(K.cpu().numpy(), g["Ksq"]) < 1e-12
    for k in KEYS:
        assert relerr(dK[k].cpu().numpy(), g[f"dKsq_{k}"]) < 1e-11, k
    # square, no derivatives (lower-tile fast path + mirror)
    K0 = gp.acosker(th, X, X, C=C, dC=None, diag=False)
    assert relerr(K0.cpu().numpy(), g["Ksq"]) < 1e-12
    assert torch.equal(K0, K0.T)
    # rectangular
    K, dK = gp.acosker(th, X, X2, C=C, dC=dC, diag=False)
    assert relerr(K.cpu().numpy(), g["Krc"]) < 1e-12
    for k in KEYS:
        assert relerr(dK[k].cpu().numpy(), g[f"dKrc_{k}"]) < 1e-11, k
    # one row (the predict loop's shape)
    assert relerr(gp.acosker(th, X1, X, C=C).cpu().numpy(), g["K1"]) < 1e-12
    # diagonal
    Kv, dKv = gp.acosker(th, X, x2=None, C=C, dC=dC, diag=True)
    assert relerr(Kv.cpu().numpy(), g["Kv"]) < 1e-13
    for k in KEYS:
        assert relerr(dKv[k].cpu().numpy(), g[f"dKv_{k}"]) < 1e-12, k
    # duplicate / negated rows exercise the clip at |cos| = 1
    Xd = T(g["X"]).clone()
    Xd[1] = Xd[0]
    Xd[2] = -Xd[0]
    Xd = Xd[:, mask].contiguous()
    assert relerr(gp.acosker(th, Xd, Xd, C=C).cpu().numpy(), g["Kdup"]) < 1e-12
