"""Pins the compiled C restatement of the loop (oracle/cosmo_oracle_c.c, the bench's compiled CPU baseline) against the
NumPy oracle (which is pinned on the reference's goldens): same statuses / iteration counts / rho updates, iterates to 1e-8.
Summation order of the dots and norms differs (sequential vs pairwise), so the comparison is not bitwise."""
import subprocess
import os
import numpy as np
import pytest

import cosmo_jl_amd as cj
from oracle import cosmo_oracle as O
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def OC():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    from oracle import cosmo_oracle_c
    return cosmo_oracle_c


def _pair(OC, prob, **kw):
    st = O.Settings(kkt_solver="cg", **kw)
    ws1 = O.Workspace(prob["P"], prob["q"], prob["A"], prob["b"], util.oracle_cones(prob["sets"]), st)
    ws2 = O.Workspace(prob["P"], prob["q"], prob["A"], prob["b"], util.oracle_cones(prob["sets"]), st)
    return ws1.optimize(), OC.run(ws2)


def _check(r, c, tol=1e-8):
    assert c["status"] == r.status
    assert c["iter"] == r.iter
    assert len(c["rho_updates"]) == len(r.rho_updates)
    np.testing.assert_allclose(c["rho_updates"], r.rho_updates, rtol=max(1e-7, 10 * tol))
    sc = max(1.0, float(np.max(np.abs(r.x))))
    assert np.max(np.abs(c["x"] - r.x)) <= tol * sc
    assert np.max(np.abs(c["s"] - r.s)) <= tol * max(1.0, float(np.max(np.abs(r.s))))
    assert np.max(np.abs(c["y"] - r.y)) <= tol * max(1.0, float(np.max(np.abs(r.y))))
    assert abs(c["obj_val"] - r.obj_val) <= tol * max(1.0, abs(r.obj_val))
    assert abs(c["r_prim"] - r.r_prim) <= 10 * tol * max(1.0, r.max_norm_prim)   # a difference of O(max_norm) terms
    assert abs(c["cg_iters_total"] - int(np.sum(r.cg_iters))) <= max(2, 0.01 * np.sum(r.cg_iters))


def test_c_oracle_cfg1_dense_qp(OC):
    prob = cj.problems.dense_qp()
    r, c = _pair(OC, prob, tol_constant=1e-10, tol_exponent=0.0)
    assert r.status == "Solved"
    _check(r, c)


def test_c_oracle_sparse_box_qp_default_cg(OC):
    prob = cj.problems.sparse_box_qp(n=2000, m=4000, nnz=40000, seed=5)
    r, c = _pair(OC, prob, max_iter=300)
    _check(r, c, tol=1e-6)          # inexact CG: last-bit differences move the stopping iteration of single solves


def test_c_oracle_mixed_zero_nonneg_box(OC):
    rng = np.random.default_rng(11)
    prob = util.random_qp(rng, 60, 10, 30, 40)      # equality / loose / one-sided box rows: all three rho classes
    r, c = _pair(OC, prob, tol_constant=1e-10, tol_exponent=0.0, max_iter=500)
    assert r.status == "Solved"
    _check(r, c, tol=1e-5)          # CG runs into its maxiter = n here (unconverged solves), which amplifies rounding differences


def test_c_oracle_zero_nonneg_unscaled(OC):
    rng = np.random.default_rng(12)
    prob = util.random_qp(rng, 80, 10, 60, 0)
    r, c = _pair(OC, prob, scaling=0, tol_constant=1e-10, tol_exponent=0.0, max_iter=500)
    _check(r, c, tol=1e-7)


def test_c_oracle_max_iter_status(OC):
    prob = cj.problems.dense_qp(n=50, half_m=40, seed=3)
    r, c = _pair(OC, prob, max_iter=25, eps_abs=0.0, eps_rel=0.0, tol_constant=1e-10, tol_exponent=0.0)
    assert r.status == "Max_iter_reached"
    _check(r, c)
