#!/usr/bin/env python
"""CPU study: would a Jacobi-preconditioned CG cut the Krylov work of the reduced KKT solves?

The reference runs IterativeSolvers' cg! WITHOUT a preconditioner (src/linear_solver/kktsolver_indirect.jl:70); the literal recurrence is
and stays the default of the device library.  This script replays the reduced solves of the first ADMM iterations of a BASELINE
configuration (same operator L = P + sigma I + A' rho A, right-hand side, warm start and the same stopping rule ||r||_2 <= tol_k / ||rhs||
on the TRUE residual) through (a) the oracle's restated cg! and (b) the textbook preconditioned CG with M = diag(L):
    d_j = P_jj + sigma + sum_i rho_i a_ij^2.
Only if (b) needs >= 3x fewer iterations would an opt-in `kkt_kind` be worth building (it would change every iterate, so it could never be
the parity path).

usage: pcg_jacobi_count.py cfg2|cfg5|cfg5small [n_admm_iterations=3]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                      # noqa: E402
from oracle import cosmo_oracle as O    # noqa: E402
from cosmo_jl_amd import problems       # noqa: E402
from tests.util import oracle_cones     # noqa: E402


def pcg_jacobi(x0, mul, b, dinv, abstol, maxiter):
    x = x0.copy()
    r = b - mul(x)
    z = dinv * r
    p = z.copy()
    rz = float(r @ z)
    k = 0
    while np.linalg.norm(r) > abstol and k < maxiter:
        Lp = mul(p)
        a = rz / float(p @ Lp)
        x += a * p
        r -= a * Lp
        z = dinv * r
        rz_new = float(r @ z)
        p = z + (rz_new / rz) * p
        rz = rz_new
        k += 1
    return k


def main(which, iters):
    if which == "cfg2":
        prob = problems.sparse_box_qp()
    elif which == "cfg5":
        prob = problems.chordal_sdp()
    else:
        prob = problems.chordal_sdp(ncliques=40, n_total=6000, n_zero=100, n_nonneg=500)
    st = O.Settings(max_iter=iters, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9, kkt_solver="cg", adaptive_rho=True)
    ws = O.Workspace(prob["P"], prob["q"], prob["A"], prob["b"], oracle_cones(prob["sets"]), st)
    rows = []
    orig = O.cg_v09
    A2 = ws.A.copy(); A2.data = A2.data ** 2

    def spy(x, mul, b, abstol, maxiter):
        x0 = x.copy()
        it = orig(x, mul, b, abstol, maxiter)
        d = ws.P.diagonal() + ws.st.sigma + A2.T @ ws.rho_vec
        t0 = time.time()
        kp = pcg_jacobi(x0, mul, b, 1.0 / d, abstol, maxiter)
        rows.append((len(rows) + 1, it, kp, abstol, d.max() / d.min(), time.time() - t0))
        print("solve %2d: cg! (reference recurrence) %5d iterations, Jacobi-PCG %5d, abstol %.3e, diag(L) max/min %.2e (%.1f s)" % rows[-1], flush=True)
        return it

    O.cg_v09 = spy
    try:
        ws.optimize()
    finally:
        O.cg_v09 = orig
    a = sum(r[1] for r in rows); b = sum(r[2] for r in rows)
    print("%s (n=%d, m=%d): %d solves, %d (cg!) vs %d (Jacobi-PCG) Krylov iterations: ratio %.2f" % (which, prob["A"].shape[1], prob["A"].shape[0], len(rows), a, b, a / max(b, 1)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "cfg5small", int(sys.argv[2]) if len(sys.argv) > 2 else 3)
