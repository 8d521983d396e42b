"""Generates tests/golden/oracle_trajectories.npz: seeded problems run through the CPU oracle for a fixed number of iterations
(tight CG, eps = 0).  The reference itself is Julia and cannot run in this environment (no julia binary), so these are
REGRESSION fixtures of the restated algorithm, not reference outputs: they pin the oracle against accidental change (CPU test) and
give the HIP path a committed target that does not depend on the oracle code of the day (GPU test).  The reference's own literal
known answers live next to them in reference_literals.json with their file:line.  Usage:  python tests/golden/make_fixtures.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cosmo_jl_amd as cj            # noqa: E402  (problem generators only; no device code is touched)
from oracle import cosmo_oracle as O  # noqa: E402
from tests import util               # noqa: E402

CASES = {
    # name: (generator, iterations)
    "qp_mixed": (lambda: util.random_qp(np.random.default_rng(101), 50, 6, 30, 24), 120),
    "socp": (lambda: cj.problems.socp(n=40, m=80, ncones=8, nnz=600, seed=7), 100),
    "sdp_small": (lambda: util.random_qp(np.random.default_rng(103), 30, 4, 10, 0, psd_tri_dims=(5, 9), p_shift=1.0), 80),
    "box_qp": (lambda: cj.problems.sparse_box_qp(n=300, m=600, nnz=6000, seed=9), 90),
}
SETTINGS = dict(kkt_solver="cg", tol_constant=1e-10, tol_exponent=0.0, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9)


def run_case(name):
    gen, iters = CASES[name]
    p = gen()
    ws = O.Workspace(p["P"], p["q"], p["A"], p["b"], util.oracle_cones(p["sets"]), O.Settings(max_iter=iters, **SETTINGS))
    r = ws.optimize()
    return p, r


if __name__ == "__main__":
    out = {}
    for name in CASES:
        p, r = run_case(name)
        out[name + "/x"] = r.x; out[name + "/s"] = r.s; out[name + "/y"] = r.y
        out[name + "/rho_updates"] = np.array(r.rho_updates)
        out[name + "/scalars"] = np.array([r.iter, r.r_prim, r.r_dual, r.obj_val, float(np.sum(r.cg_iters))])
        print(name, r.status, r.iter, len(r.rho_updates), "%.6e" % r.obj_val)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_trajectories.npz"), **out)
