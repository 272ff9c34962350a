"""The block subspace eigensolver of the truncated regime (gaussian_processes_amd/eigtop.py) is written against two
callables, a GEMM and a Cholesky-with-inverse; here it runs on torch CPU stand-ins for them, so its logic (block
growth, sweep prediction, residual certificate, refusals, determinism, sign convention) is covered without a GPU.
The GPU suite runs the same function on the library's own primitives (tests/test_gpu_dropin.py)."""
import numpy as np
import torch

from gaussian_processes_amd import eigtop


def cpu_matmul(A, B, transA=False, transB=False):
    return (A.T if transA else A) @ (B.T if transB else B)


def cpu_cholesky(M, want_inverse=False):
    L, info = torch.linalg.cholesky_ex(M)
    if int(info) != 0:
        return None, None, 0.0, int(info)
    Li = torch.linalg.solve_triangular(L, torch.eye(M.shape[0], dtype=M.dtype), upper=False)
    return L, Li, float(2 * torch.log(torch.diagonal(L)).sum()), 0


def kernel_like_matrix(n, seed=0):
    """SPD matrix with a spectrum like the arc-cosine kernel matrices of the fit: a few large eigenvalues, a
    smooth decay through the threshold, no gap at it."""
    g = torch.Generator().manual_seed(seed)
    Q, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64, generator=g))
    lam = 1e5 * (1.0 + torch.arange(n, dtype=torch.float64)) ** -2.2 + 1e-3
    return (Q * lam) @ Q.T, lam, Q


def test_subspace_solver_matches_eigh_on_cpu_primitives():
    n, tol = 900, 1e-4
    K, lam, Q = kernel_like_matrix(n)
    K = (K + K.T) / 2
    w, U = torch.linalg.eigh(K)
    keep = w > max(float(w[-1]) * tol, tol)
    out = eigtop.top_eigenpairs(K, tol, cpu_matmul, cpu_cholesky, k0=128)
    assert out is not None
    vals, vecs, info = out
    assert 10 < int(keep.sum()) < n // 3 and vals.shape[0] == int(keep.sum())
    assert float(((vals - w[keep]).abs() / w[keep]).max()) < 1e-10
    P = U[:, keep].T @ vecs
    eye = torch.eye(P.shape[0], dtype=torch.float64)
    assert float((P.T @ P - eye).abs().max()) < 1e-9
    assert float((vecs.T @ vecs - eye).abs().max()) < 1e-12
    assert bool((vecs.abs().max(0).values == vecs.max(0).values).all())         # largest component positive
    again = eigtop.top_eigenpairs(K, tol, cpu_matmul, cpu_cholesky, k0=128)
    assert torch.equal(again[0], vals) and torch.equal(again[1], vecs)           # deterministic in K
    assert info["k"] >= vals.shape[0] and info["rr"] >= 1


def test_subspace_solver_grows_its_block_and_declines_when_too_much_is_kept():
    n = 900
    K, lam, Q = kernel_like_matrix(n, seed=1)
    K = (K + K.T) / 2
    w = torch.linalg.eigvalsh(K)
    # a start block smaller than the kept count: the solver has to grow it
    tol = 1e-4
    n_keep = int((w > max(float(w[-1]) * tol, tol)).sum())
    out = eigtop.top_eigenpairs(K, tol, cpu_matmul, cpu_cholesky, k0=max(8, n_keep // 2))
    assert out is not None and out[0].shape[0] == n_keep and out[2]["grown"] >= 1
    # a rule that keeps more than a third of the spectrum is left to the full eigendecomposition
    tol = 1e-8
    assert int((w > max(float(w[-1]) * tol, tol)).sum()) > n // 3
    assert eigtop.top_eigenpairs(K, tol, cpu_matmul, cpu_cholesky, k0=128) is None


def test_subspace_solver_declines_an_ambiguous_count():
    """An eigenvalue sitting on the threshold within its residual: the count cannot be certified."""
    n = 600
    g = torch.Generator().manual_seed(2)
    Q, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64, generator=g))
    lam = 1e4 * (1.0 + torch.arange(n, dtype=torch.float64)) ** -2.0 + 1e-3
    tol = 1e-3
    lam[40] = float(lam[0]) * tol * (1 + 1e-14)                  # exactly at lambda_max * tol (to rounding)
    K = (Q * lam) @ Q.T
    K = (K + K.T) / 2
    out = eigtop.top_eigenpairs(K, tol, cpu_matmul, cpu_cholesky, k0=128, max_sweeps=24)
    w = torch.linalg.eigvalsh(K)
    if out is not None:      # rounding put it clearly on one side: then the count must agree with eigh's
        assert out[0].shape[0] == int((w > max(float(w[-1]) * tol, tol)).sum())


def test_kept_eigenspace_without_a_dense_eigendecomposition():
    """basis="subspace": the same sweeps, then the spectral projector of the k x k Rayleigh quotient matrix by a
    Cayley transform and the Newton-Schulz sign iteration instead of its eigendecomposition -- same count as eigh's,
    the same space to rounding, a dense K~_b = B^T K B whose eigenvalues are the kept ones, and a basis that depends on
    the space alone (two different block sizes give the same columns)."""
    n, tol = 900, 1e-4
    K, lam, Q = kernel_like_matrix(n)
    K = (K + K.T) / 2
    w, U = torch.linalg.eigh(K)
    keep = w > max(float(w[-1]) * tol, tol)
    outs = []
    for k0 in (128, 160):
        out = eigtop.top_eigenpairs(K, tol, cpu_matmul, cpu_cholesky, k0=k0, basis="subspace")
        assert out is not None and out[0] is None and out[2]["route"] == "subspace" and out[2]["rr"] == 0
        B, info = out[1], out[2]
        nk = int(keep.sum())
        assert B.shape == (n, nk) and info["n"] == nk
        eye = torch.eye(nk, dtype=torch.float64)
        P = U[:, keep].T @ B
        assert float((P.T @ P - eye).abs().max()) < 1e-9 and float((B.T @ B - eye).abs().max()) < 1e-12
        assert float(((torch.linalg.eigvalsh(info["K_tilde_b"]) - w[keep]).abs() / w[keep]).max()) < 1e-10
        assert float((info["K_tilde_inv_b"] @ info["K_tilde_b"] - eye).abs().max()) < 1e-8
        outs.append(B)
    assert float((outs[0] - outs[1]).abs().max()) < 1e-6
    # an eigenvalue on the threshold: the sign iteration cannot settle, the eigenpair route takes over (and may decline too)
    g = torch.Generator().manual_seed(2)
    Qm, _ = torch.linalg.qr(torch.randn(600, 600, dtype=torch.float64, generator=g))
    lam = 1e4 * (1.0 + torch.arange(600, dtype=torch.float64)) ** -2.0 + 1e-3
    lam[40] = float(lam[0]) * 1e-3 * (1 + 1e-14)
    Km = (Qm * lam) @ Qm.T
    out = eigtop.top_eigenpairs((Km + Km.T) / 2, 1e-3, cpu_matmul, cpu_cholesky, k0=128, max_sweeps=24, basis="subspace")
    assert out is None or out[2].get("route") != "subspace"


def test_scaled_sign_iteration_needs_fewer_steps_for_the_same_projector():
    """The Newton-Schulz sign iteration of ``_kept_subspace`` with the Chen-Chow scaling (every step applied to c X,
    c^2 = 3 / (1 + l + l^2) for an assumed lower bound l of |x|) against the plain iteration: the same count and basis,
    in fewer steps -- also when the nearest Ritz value is CLOSER to the threshold than the assumed bound (the scaling
    is then merely not optimal)."""
    g = torch.Generator().manual_seed(4)
    k, tol = 192, 1e-3
    Z, _ = torch.linalg.qr(torch.randn(k, k, dtype=torch.float64, generator=g))
    for nearest in (3e-2, 1e-5):                       # relative distance of the nearest eigenvalue from tau
        theta = torch.logspace(0, -4.5, k, dtype=torch.float64)
        tau = float(theta[0]) * tol
        j = int(torch.argmin((theta - tau).abs()))
        theta[j] = tau * (1 + nearest)
        apart = ((theta - tau).abs() / tau >= nearest * 0.999)
        theta = torch.where(apart, theta, tau * (1 + 2 * nearest) * torch.ones_like(theta))
        S = (Z * theta) @ Z.T
        S = (S + S.T) / 2
        Q = torch.eye(k, dtype=torch.float64)
        res = {}
        for scaled in (False, True):
            out = eigtop._kept_subspace(Q, S.clone(), S, tol, 0.0, cpu_matmul, cpu_cholesky, 1e-7, scaled=scaled)
            assert out is not None
            res[scaled] = out
        nk = int((theta > tau).sum())
        assert res[True]["n"] == res[False]["n"] == nk
        assert res[True]["sign_iterations"] < res[False]["sign_iterations"], (nearest, res[True]["sign_iterations"],
                                                                                 res[False]["sign_iterations"])
        assert float((res[True]["B"] - res[False]["B"]).abs().max()) < 1e-7 / max(nearest, 1e-3) * 1e-3
        top = Z[:, theta > tau]
        P = top.T @ res[True]["B"]
        assert float((P.T @ P - torch.eye(nk, dtype=torch.float64)).abs().max()) < 1e-8


def test_dense_projector_route_for_small_matrices():
    """``kept_eigenspace_dense``: the kept eigenspace from the spectral projector of K itself (no sweeps) -- the count of
    eigh, the same space to rounding, the same canonical basis as the sweeps route returns for it (also when the size
    is not a multiple of the GEMM's K step and K is padded), dense K~_b with the kept eigenvalues; everything kept or an
    eigenvalue on the threshold: declines."""
    tol = 1e-4
    for n in (900, 512):
        K, lam, Q = kernel_like_matrix(n)
        K = (K + K.T) / 2
        w, U = torch.linalg.eigh(K)
        keep = w > max(float(w[-1]) * tol, tol)
        nk = int(keep.sum())
        out = eigtop.kept_eigenspace_dense(K, tol, cpu_matmul, cpu_cholesky)
        assert out is not None and out[0] is None
        B, info = out[1], out[2]
        assert B.shape == (n, nk) and info["n"] == nk and info["route"] == "subspace" and info["sweeps"] == 0
        assert "state" not in info
        eye = torch.eye(nk, dtype=torch.float64)
        P = U[:, keep].T @ B
        assert float((P.T @ P - eye).abs().max()) < 1e-10 and float((B.T @ B - eye).abs().max()) < 1e-12
        assert float(((torch.linalg.eigvalsh(info["K_tilde_b"]) - w[keep]).abs() / w[keep]).max()) < 1e-10
        assert float((info["K_tilde_inv_b"] @ info["K_tilde_b"] - eye).abs().max()) < 1e-8
        assert abs(info["lam_max"] - float(w[-1])) <= 1e-12 * float(w[-1])
        sw = eigtop.top_eigenpairs(K, tol, cpu_matmul, cpu_cholesky, k0=160, basis="subspace")
        assert sw is not None and sw[0] is None
        assert float((sw[1] - B).abs().max()) < 1e-6
    # a top eigenvalue that is NOT well separated: the power iteration for lambda_max runs until the Rayleigh quotient
    # stands still (tau = lambda_max * tol decides the count)
    g = torch.Generator().manual_seed(5)
    Qm, _ = torch.linalg.qr(torch.randn(300, 300, dtype=torch.float64, generator=g))
    lam = torch.logspace(0, -7, 300, dtype=torch.float64)
    lam[1] = 0.97
    Km = (Qm * lam) @ Qm.T
    out = eigtop.kept_eigenspace_dense((Km + Km.T) / 2, 1e-3, cpu_matmul, cpu_cholesky)
    assert out is not None and out[2]["n"] == int((lam > 1e-3).sum()) and abs(out[2]["lam_max"] - 1.0) < 1e-10
    # nothing dropped: reported as such (an exact all-kept proof); an eigenvalue on the threshold: declines
    for nf in (300, 304):                                   # padded inside / a multiple of the K step as it is
        Qf, _ = torch.linalg.qr(torch.randn(nf, nf, dtype=torch.float64, generator=g))
        Kf = (Qf * torch.linspace(1.0, 2.0, nf, dtype=torch.float64)) @ Qf.T
        allk = eigtop.kept_eigenspace_dense((Kf + Kf.T) / 2, 1e-4, cpu_matmul, cpu_cholesky)
        assert allk is not None and allk[1] is None and allk[2]["all_kept"] and allk[2]["n"] == nf
    lam2 = torch.logspace(0, -7, 300, dtype=torch.float64)
    lam2[40] = 1e-3 * (1 + 1e-14)
    Ka = (Qm * lam2) @ Qm.T
    assert eigtop.kept_eigenspace_dense((Ka + Ka.T) / 2, 1e-3, cpu_matmul, cpu_cholesky) is None


def test_warm_start_from_a_nearby_matrix_needs_fewer_sweeps_and_lands_on_the_same_space():
    """``start=info["state"]`` of a solve on K, handed to the solve of a perturbed K (what varGP does from one EM
    iteration to the next): fewer sweeps for the same certificate, the same count as eigh and the same space to rounding,
    the same canonical basis as the cold solve of that matrix to the certificate's accuracy."""
    n, tol = 900, 1e-4
    K, lam, Q = kernel_like_matrix(n)
    K = (K + K.T) / 2
    cold0 = eigtop.top_eigenpairs(K, tol, cpu_matmul, cpu_cholesky, k0=160, basis="subspace")
    assert cold0 is not None and cold0[2]["warm"] is False and "state" in cold0[2]
    g = torch.Generator().manual_seed(9)
    E = torch.randn(n, n, dtype=torch.float64, generator=g)
    E = (E + E.T) / 2
    w = torch.linalg.eigvalsh(K)
    K2 = K * (1 + 1e-4) + 1e-4 * tol * float(w[-1]) * E / float(torch.linalg.matrix_norm(E, 2))
    w2, U2 = torch.linalg.eigh(K2)
    keep2 = w2 > max(float(w2[-1]) * tol, tol)
    cold = eigtop.top_eigenpairs(K2, tol, cpu_matmul, cpu_cholesky, k0=160, basis="subspace")
    warm = eigtop.top_eigenpairs(K2, tol, cpu_matmul, cpu_cholesky, k0=160, basis="subspace", start=cold0[2]["state"])
    assert warm is not None and warm[2]["warm"] is True and warm[2]["route"] == "subspace"
    assert warm[2]["sweeps"] < cold[2]["sweeps"] and warm[2]["start_angle"] < 1e-2
    assert warm[2]["angle"] <= 1e-7 and warm[2]["n"] == int(keep2.sum()) == cold[2]["n"]
    P = U2[:, keep2].T @ warm[1]
    assert float((P.T @ P - torch.eye(P.shape[1], dtype=torch.float64)).abs().max()) < 1e-9
    assert float((warm[1] - cold[1]).abs().max()) < 1e-5            # the canonical basis, up to the certificate
    # a state that does not fit (another size) is ignored: cold solve
    other = eigtop.top_eigenpairs(K2[:800, :800].contiguous(), tol, cpu_matmul, cpu_cholesky, k0=160, basis="subspace", start=cold0[2]["state"])
    assert other is None or other[2]["warm"] is False
