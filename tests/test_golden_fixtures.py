"""Committed fixtures under tests/golden/: (1) the reference's literal known answers (reference_literals.json, file:line inside)
replayed on the oracle [CPU] and on the device library [GPU]; (2) seeded regression trajectories of the oracle
(oracle_trajectories.npz, made by tests/golden/make_fixtures.py) reproduced by today's oracle [CPU] and matched by the HIP path
through the C ABI [GPU] at the trajectory tolerance of SURVEY 8c (tight CG: 1e-7 relative)."""
import importlib.util
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

import cosmo_jl_amd as cj
from oracle import cosmo_oracle as O
from tests import util

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LIT = json.load(open(os.path.join(HERE, "reference_literals.json")))
FIX = np.load(os.path.join(HERE, "oracle_trajectories.npz"))
_spec = importlib.util.spec_from_file_location("make_fixtures", os.path.join(HERE, "make_fixtures.py"))
MK = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(MK)


def _simple():
    g = LIT["simple_qp"]
    return np.array(g["P"]), np.array(g["q"]), np.array(g["A"]), np.array(g["l"]), np.array(g["u"]), g


def test_reference_literal_simple_qp_on_the_oracle():
    P, q, A, l, u, g = _simple()
    Ai, bi, cones = O.assemble([O.Constraint(A, np.zeros(3), O.Box(l, u))])
    r = O.solve(P, q, Ai, bi, cones)
    assert r.status == g["status"] and np.linalg.norm(r.x - g["x"]) < g["tol_x"] and abs(r.obj_val - g["obj"]) < g["tol_obj"]


@pytest.mark.parametrize("name", sorted(MK.CASES))
def test_oracle_reproduces_its_committed_trajectories(name):
    _, r = MK.run_case(name)
    for key, val in (("x", r.x), ("s", r.s), ("y", r.y)):
        ref = FIX[name + "/" + key]
        assert np.max(np.abs(val - ref)) <= 1e-9 * max(1.0, float(np.max(np.abs(ref)))), (name, key)   # same code, other BLAS builds
    sc = FIX[name + "/scalars"]
    assert r.iter == int(sc[0]) and len(r.rho_updates) == FIX[name + "/rho_updates"].size
    assert abs(float(np.sum(r.cg_iters)) - sc[4]) <= 0.01 * sc[4] + 2


@pytest.mark.gpu
def test_reference_literal_simple_qp_on_the_device():
    P, q, A, l, u, g = _simple()
    md = cj.Model()
    cj.assemble(md, P, q, [cj.Constraint(A, np.zeros(3), cj.Box(l, u))], settings=cj.Settings())
    r = cj.optimize(md)
    assert r.status == g["status"] and np.linalg.norm(r.x - g["x"]) < g["tol_x"] and abs(r.obj_val - g["obj"]) < g["tol_obj"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MK.CASES))
def test_device_matches_committed_trajectories(name):
    gen, iters = MK.CASES[name]
    p = gen()
    st = cj.Settings(kkt_solver=cj.with_options(cj.CGIndirectKKTSolver, tol_constant=1e-10, tol_exponent=0.0), max_iter=iters, eps_abs=0.0, eps_rel=0.0,
                     check_infeasibility=10 ** 9)
    md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], st)
    r = cj.optimize(md)
    sc = FIX[name + "/scalars"]
    assert r.iter == int(sc[0]) and r.status == "Max_iter_reached"
    for key, val in (("x", r.x), ("s", r.s), ("y", r.y)):
        ref = FIX[name + "/" + key]
        assert np.max(np.abs(val - ref)) <= 1e-7 * max(1.0, float(np.max(np.abs(ref)))), (name, key)
    assert len(r.info.rho_updates) == FIX[name + "/rho_updates"].size
    # rho_new = rho sqrt(ratio of normalised residuals): once a problem has converged (residuals ~1e-11) that ratio is rounding noise
    assert np.allclose(r.info.rho_updates, FIX[name + "/rho_updates"], rtol=1e-3)
    assert abs(r.obj_val - sc[3]) <= 1e-6 * (1 + abs(sc[3]))
