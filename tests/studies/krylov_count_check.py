#!/usr/bin/env python
"""External confirmation of the Krylov work count of BASELINE config 2 (CPU only).

The headline benchmark is ~99.8 % CG iterations, and `mean_cg_iters_per_admm_iter` (K-bar ~ 480-560) comes from our restatement of
IterativeSolvers' cg! (not vendored in the reference).  This script runs the first ADMM iterations of the FULL-SIZE cfg2 instance on
the oracle and replays every reduced solve (same operator, right-hand side, warm start, absolute tolerance tol_k / ||rhs||,
kktsolver_indirect.jl:70) through SciPy's independently written cg: per-solve iteration counts side by side.

usage: krylov_count_check.py [n_admm_iterations=4]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                      # noqa: E402
import scipy.sparse.linalg as spla      # noqa: E402
from oracle import cosmo_oracle as O    # noqa: E402
from cosmo_jl_amd import problems       # noqa: E402
from tests.util import oracle_cones     # noqa: E402


def main(iters):
    prob = problems.sparse_box_qp()
    st = O.Settings(max_iter=iters, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9, kkt_solver="cg", adaptive_rho=True)
    ws = O.Workspace(prob["P"], prob["q"], prob["A"], prob["b"], oracle_cones(prob["sets"]), st)
    rows = []
    orig = O.cg_v09

    def spy(x, mul, b, abstol, maxiter):
        x0 = x.copy()
        t0 = time.time()
        it = orig(x, mul, b, abstol, maxiter)
        t1 = time.time()
        nn = len(b)
        cnt = [0]
        spla.cg(spla.LinearOperator((nn, nn), matvec=mul, dtype=np.float64), b, x0=x0, rtol=0.0, atol=abstol, maxiter=nn,
                callback=lambda xk: cnt.__setitem__(0, cnt[0] + 1))
        rows.append((len(rows) + 1, it, cnt[0], abstol, t1 - t0))
        print("solve %2d: oracle cg_v09 %4d iterations, scipy cg %4d, abstol %.3e (%.1f s)" % rows[-1], flush=True)
        return it

    O.cg_v09 = spy
    try:
        ws.optimize()
    finally:
        O.cg_v09 = orig
    a = sum(r[1] for r in rows)
    b = sum(r[2] for r in rows)
    print("cfg2 full size (n=%d, m=%d): %d solves, total %d (oracle) vs %d (scipy) Krylov iterations; mean per solve %.1f vs %.1f"
          % (prob["A"].shape[1], prob["A"].shape[0], len(rows), a, b, a / len(rows), b / len(rows)))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
