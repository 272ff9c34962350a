"""Leading eigenpairs of the kernel matrix without a full eigendecomposition.

The reference's eigen-stabilisation (``utils.py:1682-1694``, ``1808-1818``) keeps the eigenvectors of K~
with ``lambda > max(lambda_max * EIGVAL_TOL, EIGVAL_TOL)`` -- about 520 of 8192 at the default tolerance
on the bench inputs -- yet pays for all N of them (``torch.linalg.eigh``: 0.37 s at N = 8192, once per EM
iteration, most of a default-tolerance fit).  Here the wanted ones come from block subspace iteration with a
Rayleigh-Ritz step, built from the library's own MFMA GEMM and Cholesky:

    Q <- orth(K~ Q)   (CholeskyQR2: G = Y^T Y, G = L L^T, Q = Y L^-T, twice)      ... a few times
    S = Q^T K~ Q,  S = Z diag(theta) Z^T   (k x k, k ~ 1024: the only dense eigenproblem left)
    X = Q Z,  residuals ||K~ x_i - theta_i x_i||

The block is grown until its smallest Ritz value lies well below the threshold (so that the iteration
contracts every kept direction by at least ``lambda_{k+1} / tau`` per step) and iterated until, for every kept
pair, ``residual / (theta_i - theta_min(block))`` is below ``angle_tol``: a bound on how far each kept
eigenvector can stick out of the block's span (the spectrum outside the block lies below its smallest Ritz
value once the block has converged) -- a stopping criterion, NOT a certificate for the kept subspace itself,
whose conditioning is set by the gap between the last kept and the first dropped eigenvalue (the reference's own
``eigh`` + truncation has the same sensitivity there).  That second quantity is reported as
``info['angle_kept_vs_dropped']`` (residual over the gap across the threshold) and the count is only accepted when
no Ritz value sits within its residual of the threshold.  Anything else (slow convergence, ambiguous count, block
larger than half of N, Cholesky failure) returns ``None`` and the caller takes the full ``eigh`` route; parity
of this route with the ``eigh`` route is tested end to end (``tests/test_gpu_dropin.py``), not pinned to a
reference fixture at N >= 4096.  The start block comes from a fixed seed, so the result is a deterministic
function of K~ -- ``test(at_iteration)`` rebuilds exactly the basis the tracked ``(m_b, V_b)`` were expressed in --
and every eigenvector is signed so that its largest component is positive.

Tried and dropped (round 3, profiles/r03_whole_fit_breakdown.json): starting the block from the previous EM
iteration's kept eigenvectors.  The sweeps a default-tolerance fit needs are set by the directions just above the
threshold, which contract by lambda_{k+1} / tau per sweep whatever the start, and a block of n_kept + 25 %
columns does not reach below tau / 2, so it is grown anyway: 0.62 s per fit against 0.44 s from the seeded start.

Round 4 (``basis="subspace"``, the default route of ``utils._stabilised_basis``):
  * Chebyshev-shifted sweeps (9 instead of 16 for the same certificate);
  * ``_kept_subspace``: count and basis from the spectral projector of the k x k Rayleigh quotient matrix (Cayley
    transform + scaled Newton-Schulz sign iteration) instead of its eigendecomposition, ``B`` the canonical orthonormal
    basis of the kept eigenspace;
  * ``start=``: the WHOLE converged block of the previous EM iteration (not its kept eigenvectors, as tried above) with
    the sweeps planned from the measured distance -- this one pays: 3-9 sweeps instead of 13;
  * ``kept_eigenspace_dense``: for matrices below ~1800 rows, where the kept count is a third of the matrix or more,
    the same projector step on K itself (no sweeps), lambda_max certified by a Cholesky test.
"""
from __future__ import annotations

import math

import torch


def _cholqr(Y, matmul, cholesky, rounds):
    """Orthonormal basis of range(Y) by Cholesky QR (G = Y^T Y = L L^T, Q = Y L^-T); two rounds give
    orthogonality to rounding (CholeskyQR2), one keeps an iteration basis well conditioned.  None if a Gram
    matrix is not numerically positive definite."""
    for _ in range(rounds):
        G = matmul(Y, Y, transA=True)
        G = (G + G.T) * 0.5
        L, Li, _, info = cholesky(G, want_inverse=True)
        if info != 0:
            return None
        Y = matmul(Y, Li, transB=True)          # Y L^-T
    return Y


def _chebyshev_shifts(a, m):
    """The m roots of the Chebyshev polynomial of the interval [0, a], largest and smallest alternating.  One sweep
    with each of them as its shift applies (K - s_1) ... (K - s_m), the degree-m polynomial that is smallest on
    [0, a] for its growth outside: a kept direction (lambda >= tau > a) gains 1 / T_m((2 tau - a) / a) on everything
    left outside the block, against (a / tau)^m for m unshifted sweeps.  Every factor is a contraction on its own
    (|lambda - s| <= a < tau - s for lambda in [0, a], as long as a < tau / 2 ... 0.6 tau), so the order only matters
    for how evenly the gain arrives."""
    import math
    roots = [0.5 * a * (1.0 + math.cos((2 * j - 1) * math.pi / (2 * m))) for j in range(1, m + 1)]
    order, lo, hi = [], 0, m - 1
    while lo <= hi:
        order.append(roots[lo]); lo += 1
        if lo <= hi:
            order.append(roots[hi]); hi -= 1
    return order


_HASH_CACHE = {}


def _hash_matrix(rows, cols, device, dtype):
    """(cached per shape: every basis build of a fit asks for the same one or two; 0.1 ms of small kernels otherwise)"""
    key = (rows, cols, str(device), dtype)
    hit = _HASH_CACHE.get(key)
    if hit is None:
        if len(_HASH_CACHE) >= 8:
            _HASH_CACHE.clear()
        hit = _HASH_CACHE[key] = _hash_matrix_build(rows, cols, device, dtype)
    return hit


def _hash_matrix_build(rows, cols, device, dtype):
    """Fixed pseudo-random test matrix with entries in (-1, 1): an integer hash of (row, column), evaluated on the
    device -- the same numbers on every library / torch version (a seeded generator's stream is not promised to be),
    which is what makes the basis below a function of the subspace alone."""
    i = torch.arange(rows, device=device, dtype=torch.int64)[:, None]
    j = torch.arange(cols, device=device, dtype=torch.int64)[None, :]
    x = (i * 0x9E3779B1 + j * 0x85EBCA77 + 0x165667B1) & 0xFFFFFFFF
    x = x ^ (x >> 15)
    x = (x * 0x2C1B3C6D) & 0xFFFFFFFF
    x = x ^ (x >> 12)
    x = (x * 0x297A2D39) & 0xFFFFFFFF
    x = x ^ (x >> 15)
    return (x.to(dtype) * (1.0 / 2147483648.0) - 1.0).contiguous()


# assumed lower bound of |x| at the start of the scaled sign iteration (see _kept_subspace)
_SIGN_L0 = 1e-3


def _kept_subspace(Q, Y, S, tol, a_out, matmul, cholesky, angle_tol, max_sign_iterations=40, log=None, gemm_into=None,
                   scaled=True, k_true=None):
    """The kept invariant subspace WITHOUT the k x k eigendecomposition (24 ms of rocSOLVER at k = 1024: a third of a
    basis build at N = 8192, more than half at N = 4096), from GEMMs and Cholesky factorisations only.

    Given the converged block ``Q`` (orthonormal, N x k), ``Y = K Q`` and ``S = Q^T K Q``:
      * lambda_max = the top eigenvalue of S by power iteration from its best column (the sweeps have left it within
        1e-4 of the eigenvector; ten more steps square that away), tau = max(lambda_max tol, tol);
      * X_0 = (S - tau I)(S + tau I)^-1 (Cayley transform: eigenvalues (theta - tau) / (theta + tau) in (-1, 1), so the
        1e4 of dynamic range above the threshold is compressed to [0, 1) and what is left near zero is half the relative
        distance of a Ritz value from tau), then the Newton-Schulz iteration X <- X (3 I - X^2) / 2 to sign(S - tau I);
      * P = (I + X) / 2 is the spectral projector of S onto theta > tau, n = trace P the kept count;
      * B = orth(Q P Q^T Omega) = orth(P_V Omega) with the fixed N x n test matrix Omega (CholeskyQR2 on k x n
        matrices): the orthonormal basis of the kept eigenspace V that depends on V alone, not on the block it was
        found in -- ``test(at_iteration=...)`` rebuilds the basis the tracked (m_b, V_b) were expressed in;
      * K~_b = B^T K B = U^T S U (dense n x n) and its inverse by Cholesky.
    Certificate: the part of K B that leaves the block, ||(I - Q Q^T) K B||_F / (tau - a_out), a_out >= everything the
    block left outside -- the quantity the eigenpair route bounds vector by vector.  Returns None when the sign
    iteration does not settle (a Ritz value within ~1e-7 tau of the threshold: the count is ambiguous, as in the
    eigenpair route) or a factorisation fails; the caller then takes the eigenpair route."""
    k = S.shape[0]
    k_true = k if k_true is None else k_true      # rows of the matrix proper when S carries zero padding (Q is None)
    dev, dt = S.device, S.dtype
    eye = torch.eye(k, device=dev, dtype=dt)
    # lambda_max(S)
    v = S[:, torch.argmax(torch.diagonal(S))].clone()
    v /= torch.linalg.vector_norm(v)
    for _ in range(12):                             # k x k matrix-vector products: host plumbing, as the k x k eigh was
        w = S @ v
        v = w / torch.linalg.vector_norm(w)
    lam_max = float(v @ (S @ v))                    # (one host synchronisation for the whole power iteration)
    if Q is None:
        # S is the matrix itself, not the Rayleigh quotient matrix of a converged block: its best column is no
        # eigenvector.  Iterate until the Rayleigh quotient (error = the square of the vector's) stands still, then
        # CERTIFY it: lambda_max >= the quotient always, and a Cholesky factorisation of (1 + 1e-8) quotient I - S
        # succeeds only if every eigenvalue lies below that -- so tau is the reference's to 1e-8 relative, or the route
        # declines (a start vector nearly orthogonal to the top eigenvector makes the quotient stall at lambda_2 first;
        # the certificate catches it, a power of S applied to the vector -- six normalised squarings -- gets it unstuck).
        def settle(apply, lam0, blocks):
            nonlocal v
            lam_ = lam0
            for _ in range(blocks):
                for _ in range(8):
                    w_ = apply(v)
                    v = w_ / torch.linalg.vector_norm(w_)
                nxt = float(v @ (S @ v))
                still = abs(nxt - lam_) <= 1e-13 * abs(nxt)
                lam_ = nxt
                if still:
                    break
            return lam_

        def certified(lam_):
            _, _, _, info_c = cholesky((1.0 + 1e-8) * lam_ * eye - S)
            return info_c == 0

        lam_max = settle(lambda x: S @ x, lam_max, 8)
        if not certified(lam_max):
            M = S / torch.linalg.matrix_norm(S)
            for _ in range(6):
                M = matmul(M, M)
                M = M / torch.linalg.matrix_norm(M)
            lam_max = settle(lambda x: M @ x, lam_max, 40)
            if not certified(lam_max):
                return None
    tau = max(lam_max * tol, tol)
    if not (a_out < 0.55 * tau):
        return None                                 # the block does not reach well below the threshold: eigenpair route
    L, Li, _, info = cholesky(S + tau * eye, want_inverse=True)
    if info != 0:
        return None
    Sinv = matmul(Li, Li, transA=True)
    X = matmul(S - tau * eye, Sinv)
    X = (X + X.T) * 0.5
    # Newton-Schulz, X <- X (3 I - X^2) / 2: every eigenvalue x of X grows by 3/2 per step while small and then converges
    # cubically to +-1; the smallest |x_0| is half the relative distance of the nearest Ritz value from tau (1e-2 to
    # 1e-5 on these spectra: 21-28 plain steps).  Scaled (Chen & Chow 2014): with every |x| in [l, 1], the step applied to
    # c X, c^2 = 3 / (1 + l + l^2), maps l and 1 to the same value l' = c l (3 - c^2 l^2) / 2 and everything between
    # them into [l', 1] -- small eigenvalues grow by up to 3 sqrt(3) / 2 = 2.6 per step instead of 1.5.  l is an
    # ASSUMED bound (1e-3 to start with): eigenvalues below it still grow at the scaled rate and keep their sign (the
    # map is positive on (0, sqrt 3) and c < sqrt 3), they only arrive later.  The convergence test (a host
    # synchronisation) runs once the recurrence says l' = 1; if it fails, its own figure bounds the straggler from below
    # (1 - x_min^2 <= ||X^2 - I||_F) and the scaled steps resume from there.
    its, settled = 0, False
    inplace = gemm_into is not None and not (k & 15)
    lo = _SIGN_L0 if scaled else 1.0
    while its < max_sign_iterations:
        X2 = matmul(X, X)
        if lo > 1.0 - 1e-9 and (scaled or its >= 12):
            dev2 = float(torch.linalg.matrix_norm(X2 - eye))      # Frobenius: sqrt(sum (x_i^2 - 1)^2)
            if dev2 < 1e-13 * k:
                settled = True
                break
            if scaled:
                # some |x| started below the assumed bound and is still on its way: 1 - x_min^2 <= dev2 bounds it from
                # below (rigorously, once dev2 < 1), and the scaled steps resume from that bound
                lo = math.sqrt(1.0 - dev2) if dev2 < 0.99 else 0.1
        c = math.sqrt(3.0 / (1.0 + lo + lo * lo)) if lo < 1.0 else 1.0
        if inplace:
            # X <- 1.5 c X - 0.5 c^3 X X^2 as ONE product with beta (no temporaries; powers of a symmetric matrix commute,
            # the rounding asymmetry of a step is 1e-16: symmetrised every eighth step and at the end)
            Xn = X.clone()
            gemm_into(Xn, X, X2, alpha=-0.5 * c ** 3, beta=1.5 * c)
            X = Xn
            if (its & 7) == 7:
                X = (X + X.T) * 0.5
        else:
            X = matmul(X, (1.5 * c) * eye - (0.5 * c ** 3) * X2)
            X = (X + X.T) * 0.5
        lo = min(1.0, 0.5 * c * lo * (3.0 - c * c * lo * lo))
        its += 1
    X = (X + X.T) * 0.5
    if not settled:
        return None
    tr = float(torch.trace(X))
    n = int(round(0.5 * (tr + k)))
    if abs(0.5 * (tr + k) - n) > 1e-6:
        return None
    if Q is None and n >= k_true:
        # the projector is the identity on the matrix proper: every eigenvalue lies above the (certified) threshold --
        # the exact version of the caller's all-kept proof, not a truncation
        return {"n": k_true, "all_kept": True, "tau": tau, "lam_max": lam_max, "sign_iterations": its}
    if n <= 0 or n >= k:
        return None
    P = 0.5 * (X + eye)
    if Q is None:
        M = matmul(P, _hash_matrix(k, n, dev, dt))                               # P Omega   [k, n]
    else:
        M = matmul(P, matmul(Q, _hash_matrix(Q.shape[0], n, dev, dt), transA=True))   # P Q^T Omega   [k, n]
    U = _cholqr(M, matmul, cholesky, 2)
    if U is None:
        return None
    B = U if Q is None else matmul(Q, U)
    SU = matmul(S, U)
    Ktb = matmul(U, SU, transA=True)
    Ktb = (Ktb + Ktb.T) * 0.5
    Lb, Lbi, _, info = cholesky(Ktb, want_inverse=True)
    if info != 0:
        return None
    Ktib = matmul(Lbi, Lbi, transA=True)
    Ktib = (Ktib + Ktib.T) * 0.5
    if Q is None:
        res = angle = 0.0                                                        # nothing leaves the whole space
    else:
        R_out = matmul(Y, U) - matmul(Q, SU)
        res = float(torch.linalg.matrix_norm(R_out))
        angle = res / (tau - a_out)
    if log is not None:
        log(f"eigtop: subspace route k {k} kept {n} sign iterations {its} residual leaving the block {res:.2e} angle bound {angle:.2e}")
    return {"B": B.contiguous(), "U": U, "K_tilde_b": Ktb, "K_tilde_inv_b": Ktib, "n": n, "tau": tau, "lam_max": lam_max,
            "angle": angle, "sign_iterations": its}


def kept_eigenspace_dense(K, tol, matmul, cholesky, gemm_into=None, log=None):
    """The kept eigenspace of a SMALL symmetric positive definite ``K`` (n up to ~1800) with no sweeps at all:
    ``_kept_subspace`` with the whole space as its block (Q = I, S = K) -- the spectral projector of K itself by the
    Cayley transform and the scaled sign iteration (two n^3 products per step, 12-15 steps), its trace the kept count,
    ``B = orth(P Omega)`` the same canonical basis the sweeps route returns for that space.  For matrices whose kept
    count (530-580 on the fit's kernel matrices, whatever n is) is a third or more of n a block iteration has nothing
    to discard, and rocSOLVER's ``eigh`` takes 17 / 23 / 29 / 36 ms at n = 768 / 1024 / 1280 / 1536 against 3-7 ms for
    this.  Returns ``(None, B, info)`` like ``top_eigenpairs(basis="subspace")`` (no ``state``: nothing to warm-start);
    ``(None, None, info)`` with ``info["all_kept"]`` when the projector is the identity, i.e. EVERY eigenvalue lies above
    the threshold -- the exact form of the all-kept proof the caller first attempts with norm bounds; or ``None`` when
    the count is ambiguous or a factorisation fails (the caller takes the eigh)."""
    n_true = K.shape[0]
    if n_true % 16:
        npad = (n_true + 15) // 16 * 16        # zero rows / columns: zero eigenvalues, far below any threshold
        Kp = torch.zeros((npad, npad), device=K.device, dtype=K.dtype)
        Kp[:n_true, :n_true] = K
        K = Kp
    S = (K + K.T) * 0.5
    sub = _kept_subspace(None, None, S, tol, 0.0, matmul, cholesky, 1e-7, log=log, gemm_into=gemm_into, k_true=n_true)
    if sub is None:
        return None
    if sub.get("all_kept"):
        # nothing is dropped: an exact all-kept proof (the caller's identity route), not a truncation
        return None, None, {"n": n_true, "all_kept": True, "route": "identity", "lam_max": sub["lam_max"],
                            "sign_iterations": sub["sign_iterations"]}
    B = sub["B"][:n_true].contiguous()
    info = {"k": int(S.shape[0]), "sweeps": 0, "products": 0, "grown": 0, "rr": 0, "warm": False, "angle": 0.0,
            "n": sub["n"], "route": "subspace", "dense": True, "sign_iterations": sub["sign_iterations"],
            "K_tilde_b": sub["K_tilde_b"], "K_tilde_inv_b": sub["K_tilde_inv_b"], "lam_max": sub["lam_max"]}
    return None, B, info


# what the kept directions must gain on everything the random start block left outside the block before the first
# Rayleigh-Ritz check (certificate 1e-7 with a factor three in hand; calibrated on k = 2 n and k = 1.5 n blocks of the
# fit's kernel matrices, profiles/r04_eigtop.log)
_GAIN_NEEDED = 7e7


def _filter_gain(applied, a, tau):
    """|p(tau)| / max_{[0, a]} |p| of the polynomial the sweeps so far have applied, p(x) = prod (x - s_j) (s_j = 0:
    an unshifted sweep), on a grid of [0, a]."""
    if not applied:
        return 1.0
    lam = torch.linspace(0.0, float(a), 257, dtype=torch.float64)
    p = torch.ones_like(lam)
    num = 1.0
    for sg in applied:
        p = p * (lam - sg).abs() / (tau - sg)       # ratios keep the product in range
    return 1.0 / max(float(p.max()), 1e-300)


def top_eigenpairs(K, tol, matmul, cholesky, k0=None, first_sweeps=None, max_sweeps=60, angle_tol=1e-7, seed=20240229,
                   log=None, accelerate=True, basis="eigenvectors", gemm_into=None, start=None):
    """Eigenpairs of the symmetric positive definite ``K`` with ``lambda > max(lambda_max * tol, tol)``.

    Returns ``(eigenvalues ascending [n], eigenvectors [N, n], info)`` or ``None`` when the caller should fall
    back to a full eigendecomposition.  ``matmul`` / ``cholesky`` are the library's GEMM and Cholesky wrappers
    (``utils.matmul``, ``utils.cholesky``).

    ``basis="subspace"``: the same sweeps, but the k x k eigendecomposition of the Rayleigh-Ritz step is replaced by
    ``_kept_subspace`` -- returns ``(None, B [N, n], info)`` with ``B`` the canonical orthonormal basis of the kept
    EIGENSPACE (not its eigenvectors) and ``info["K_tilde_b"]``, ``info["K_tilde_inv_b"]`` the dense n x n matrices
    ``B^T K B`` and its inverse.  Everything the fit computes downstream is invariant under the choice of an
    orthonormal basis of that space.  Falls back to the eigenpair route by itself when the
    eigh-free step declines.

    ``start`` (``basis="subspace"`` only): the ``info["state"]`` of an earlier call on a NEARBY matrix (the same fit, one
    EM iteration earlier: theta has moved a little) -- its converged block and the coordinates of its kept basis in it.
    The iteration then starts from that block, measures how far the kept space of the new matrix sticks out of it
    (the certificate's own quantity, for the old basis) and plans only the sweeps that distance needs: 4-6 instead of
    9-14 late in a fit.  The stopping criterion is the same; without ``start`` the result is a function of K alone.

    ``accelerate`` (round 4): every sweep after the first is shifted, ``Q <- orth((K - s I) Q)``, the shifts running
    through the roots of the Chebyshev polynomial of ``[0, a]``, ``a`` = the smallest Rayleigh quotient of the block
    (an upper bound of everything outside the block; available from the sweep's own product).  The spectrum of these
    kernel matrices has no gap at the threshold -- with a block of twice the kept count the first dropped-out
    eigenvalue is still tau / 4 -- so plain sweeps contract by 0.25 each (16 of them for the 1e-7 certificate);
    the shifted ones by 0.08 (9).  The orthonormalisation after EVERY sweep stays: one factor spreads the
    block over lambda_max / tau = 1e4, two would exceed what a Gram matrix in fp64 resolves."""
    import math
    N_true = K.shape[0]
    if N_true % 16:
        # the GEMM wants its inner dimension in multiples of 16: pad K~ ONCE with zero rows / columns (a zero eigenvalue
        # each, far below any threshold; the rows they add to the basis are exactly zero and are cut off again) instead
        # of having every product of every sweep pad its operands
        Np = (N_true + 15) // 16 * 16
        Kp = torch.zeros((Np, Np), device=K.device, dtype=K.dtype)
        Kp[:N_true, :N_true] = K
        out = top_eigenpairs(Kp, tol, matmul, cholesky, k0=k0, first_sweeps=first_sweeps, max_sweeps=max_sweeps, angle_tol=angle_tol,
                             seed=seed, log=log, accelerate=accelerate, basis=basis, gemm_into=gemm_into, start=start)
        if out is None:
            return None
        vals, vecs, info = out
        return vals, vecs[:N_true].contiguous(), info
    N = K.shape[0]
    dev, dt = K.device, K.dtype
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    # block: about twice the kept count of the fit's kernel matrices (500-540 at every N from 1024 up: the spectrum above
    # the threshold is set by the stimulus dimension, not by N); below N = 1792 the caller takes kept_eigenspace_dense
    # largest block worth iterating on: half of N from N = 2048 up (beyond that a dense eigendecomposition is the better
    # tool); two thirds below -- there the dense eigh (rocSOLVER: 29 / 36 / 43 ms at N = 1280 / 1536 / 1792) is slow
    # against sweeps on so small a matrix, and half of N does not reach below the threshold (the kept count stays at
    # 530-560 whatever N is)
    kmax = N // 2 if N >= 2048 else (2 * N) // 3 // 128 * 128
    k = min(k0 or min(1024, max(512, kmax // 128 * 128)), N)
    # How many sweeps before the first Rayleigh-Ritz check?  That k x k eigenproblem is the expensive step at the sizes
    # this solver serves (24 ms at k = 1024 against 3.3 ms per sweep at N = 8192), so the first check should pass.
    # Plain sweeps: a fixed 16 (8 below N = 4096), the measured need of the fit's kernel matrices.  Accelerated: planned
    # after the first (unshifted) sweep from the block's own Rayleigh quotients -- the Chebyshev gain per sweep is
    # g = t + sqrt(t^2 - 1), t = (2 tau - a) / a, and the polynomial applied must reach _GAIN_NEEDED at tau against its
    # maximum over [0, a]; ``first_sweeps`` overrides either.
    dynamic = accelerate and first_sweeps is None
    if first_sweeps is None:
        first_sweeps = (9 if accelerate else 16) if N >= 4096 else 8
    warm = None
    if (start is not None and basis == "subspace" and dynamic and tuple(start["Q"].shape) == (N, start["Q"].shape[1])
            and start["Q"].shape[1] <= kmax and start["Q"].device == dev):
        warm = start
        k = int(start["Q"].shape[1])
        Q = start["Q"]
    else:
        Q = torch.randn((N, k), generator=gen, device=dev, dtype=dt)
    done, sweeps = 0, first_sweeps
    info = {"products": 0, "grown": 0, "rr": 0, "warm": warm is not None}
    a_block = None          # upper bound of the spectrum outside the block, from the last Rayleigh-Ritz step
    a_plan = None           # the same bound as the first (unshifted) sweep of this block gave it
    warm_res = None         # warm start: ||(I - Q Q^T) K B_old||_F of the old block and kept basis
    while True:
        if k > kmax:
            return None                      # not a truncation problem any more: a full eigh is the right tool
        shifts, applied = [], []
        i = 0
        while i < sweeps:
            Y = matmul(K, Q)
            info["products"] += 1
            if warm is not None:
                # first product of a warm start (Q is still the old block): how far K's kept space sticks out of it,
                # measured on the old kept basis B = Q U:  ||(I - Q Q^T) K B||_F
                YU = matmul(Y, warm["U"])
                warm_res = float(torch.linalg.matrix_norm(YU - matmul(Q, matmul(Q, YU, transA=True))))
                warm = None
            if accelerate and (i > 0 or a_block is not None):
                if not shifts:
                    if a_block is None:
                        # Rayleigh quotients of the columns after the one unshifted sweep: max <= lambda_max (tau_est <=
                        # tau), min >= the smallest Ritz value of the block >= lambda_{k+1}, the top of the spectrum the
                        # block leaves outside.  Only an UNSHIFTED sweep gives that bound: a shifted sweep damps [0, a]
                        # inside the block too, the lowest quotients then drop below lambda_{k+1} (measured: 0.18 tau
                        # against 0.25 tau) and an interval planned from them would leave (0.18, 0.25] tau undamped.
                        rq = (Q * Y).sum(0)
                        tau_est = max(float(rq.max()) * tol, tol)
                        a = min(float(rq.min()), 0.6 * tau_est)
                        a_plan = a
                        if dynamic and a > 0:
                            t = (2.0 * tau_est - a) / a
                            g = t + math.sqrt(max(t * t - 1.0, 0.0))
                            need0 = _GAIN_NEEDED
                            if warm_res is not None:
                                # the cold start's distance is ~0.5 and needs _GAIN_NEEDED; a smaller one needs that much
                                # less (never below 30: the block's own bottom directions have to settle too)
                                angle0 = warm_res / max(tau_est - a, 1e-300)
                                info["start_angle"] = angle0
                                # (a start block further out than a random one -- theta moved a lot -- gets up to one
                                # sweep more than the cold plan rather than a second certificate pass)
                                need0 = min(8.0 * _GAIN_NEEDED, max(30.0, _GAIN_NEEDED * 2.0 * angle0))
                                warm_res = None        # (one plan per call: a grown or re-swept block is planned as before)
                            still = need0 * max(1.0, 1e-7 / angle_tol) / _filter_gain(applied, a, tau_est)
                            m = math.ceil(math.log(2.0 * max(still, 1.0)) / math.log(g))       # T_m(t) ~ g^m / 2
                            sweeps = i + int(min(max(m, 2), 24))
                    else:
                        a = a_block
                    shifts = _chebyshev_shifts(a, sweeps - i) if a > 0 else [0.0] * (sweeps - i)
                sigma = shifts.pop(0)
                if sigma != 0.0:
                    Y.sub_(Q, alpha=sigma)
                applied.append(sigma)
            else:
                applied.append(0.0)
            Q = _cholqr(Y, matmul, cholesky, 2 if i == sweeps - 1 else 1)
            if Q is None:
                return None
            i += 1
        done += sweeps
        Y = matmul(K, Q)
        info["products"] += 1
        S = matmul(Q, Y, transA=True)
        S = (S + S.T) * 0.5
        if basis == "subspace" and a_plan is not None:
            sub = _kept_subspace(Q, Y, S, tol, a_plan, matmul, cholesky, angle_tol, log=log, gemm_into=gemm_into)
            if sub is not None and 4 * sub["n"] > 3 * k and k < kmax:
                # nearly every direction of the block is kept: it does not reach below the threshold, grow it
                grow = min(max(256, (k // 4 + 127) // 128 * 128), kmax - k, N - k)
                Q = _cholqr(torch.cat([Q, torch.randn((N, grow), generator=gen, device=dev, dtype=dt)], dim=1), matmul, cholesky, 2)
                if Q is None:
                    return None
                k += grow
                info["grown"] += 1
                sweeps = first_sweeps
                a_block = a_plan = None
                continue
            if sub is not None and sub["angle"] <= angle_tol:
                info.update({"k": k, "sweeps": done, "angle": sub["angle"], "n": sub["n"], "route": "subspace",
                             "sign_iterations": sub["sign_iterations"], "K_tilde_b": sub["K_tilde_b"],
                             "K_tilde_inv_b": sub["K_tilde_inv_b"], "lam_max": sub["lam_max"],
                             "state": {"Q": Q, "U": sub["U"]}})
                return None, sub["B"], info
            if sub is not None and done < max_sweeps:
                # not converged yet: more sweeps on the same block (Chebyshev interval as planned), then again
                t = (2.0 * sub["tau"] - a_plan) / a_plan
                rho = min(0.9, 1.0 / (t + math.sqrt(max(t * t - 1.0, 0.0))))
                need = math.log(max(sub["angle"], 1e-300) / (0.3 * angle_tol)) / -math.log(rho)
                sweeps = int(min(max(2, math.ceil(need) + 1), max_sweeps - done, 24))
                a_block = a_plan
                continue
            # declined (ambiguous count, block too small, failed factorisation): the eigenpair route decides
        theta, Z = torch.linalg.eigh(S)                        # k x k, ascending
        info["rr"] += 1
        lam_max = float(theta[-1])
        tau = max(lam_max * tol, tol)
        n = int((theta > tau).sum())
        # the block must reach well below the threshold, otherwise kept directions converge slowly (and eigenvalues
        # above the threshold may still be missing from it)
        if float(theta[0]) > (0.6 if accelerate else 0.5) * tau:
            grow = min(k if not accelerate else max(256, (k // 4 + 127) // 128 * 128), N - k)
            Q = torch.cat([Q, torch.randn((N, grow), generator=gen, device=dev, dtype=dt)], dim=1)
            Q = _cholqr(Q, matmul, cholesky, 2)
            if Q is None:
                return None
            k += grow
            info["grown"] += 1
            sweeps = first_sweeps
            a_block = a_plan = None
            continue
        lo = max(0, k - n - 8)                                  # the kept pairs and a few below the threshold
        Zs = Z[:, lo:].contiguous()
        X = matmul(Q, Zs)
        R = matmul(Y, Zs) - X * theta[lo:]
        res = torch.linalg.vector_norm(R, dim=0)
        gap = theta[lo:] - theta[0]                             # distance to the spectrum outside the block (at most)
        kept = theta[lo:] > tau
        angle = float((res[kept] / gap[kept]).max()) if n > 0 else 0.0
        ambiguous = bool(((theta[lo:] - tau).abs() <= res).any())
        if log is not None:
            log(f"eigtop: k {k} sweeps {done} kept {n} angle bound {angle:.2e} theta_min/tau {float(theta[0]) / tau:.3f} ambiguous {ambiguous}")
        if angle <= angle_tol and not ambiguous:
            vals = theta[lo:][kept]
            vecs = X[:, kept]
            # deterministic sign: the largest component of every eigenvector is positive
            idx = vecs.abs().argmax(dim=0)
            sign = torch.sign(vecs[idx, torch.arange(vecs.shape[1], device=dev)])
            sign[sign == 0] = 1.0
            below = theta[lo:][~kept]
            gap_thr = float(vals[0] - below[-1]) if (n > 0 and below.numel() > 0) else float("inf")
            info.update({"k": k, "sweeps": done, "angle": angle, "n": n, "theta_min_over_tau": float(theta[0]) / tau,
                         "angle_kept_vs_dropped": float(res[kept].max()) / gap_thr if (n > 0 and gap_thr > 0) else 0.0})
            return vals.contiguous(), (vecs * sign).contiguous(), info
        if done >= max_sweeps:
            return None
        # sweeps still needed for the angle bound to pass, with two in reserve (the Ritz value overestimates
        # lambda_{k+1} a little): plain sweeps contract the kept directions by theta_min(block) / tau each, the
        # Chebyshev-shifted ones by 1 / (t + sqrt(t^2 - 1)), t = (2 tau - a) / a
        a_block = float(theta[0])
        rho = min(0.9, max(a_block / tau, 1e-3))
        if accelerate:
            t = (2.0 * tau - a_block) / a_block
            rho = min(0.9, 1.0 / (t + math.sqrt(max(t * t - 1.0, 0.0))))
        need = math.log(max(angle, 1e-300) / (0.3 * angle_tol)) / -math.log(rho) if angle > 0 else 1
        sweeps = int(min(max(2, math.ceil(need) + 2), max_sweeps - done, 24))
        Q = matmul(Q, Z)                                         # continue from the Ritz basis (same span)
