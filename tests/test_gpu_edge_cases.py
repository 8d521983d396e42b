"""GPU edge cases: empty / ragged inputs, zero-dimensional cones, long rows, tiny problems (the reference tests ragged model
shapes in test/UnitTests/model.jl, constraints.jl; here the same shapes go through the device loop and are compared with the
oracle)."""
import numpy as np
import pytest
import scipy.sparse as sp

import cosmo_jl_amd as cj
from oracle import cosmo_oracle as O
from tests import util

pytestmark = pytest.mark.gpu


def _both(P, q, A, b, sets, **st):
    model = cj.Model(); model.set(P, q, A, b, sets, cj.Settings(**st))
    res = cj.optimize(model)
    ref = O.solve(P, q, A, b, util.oracle_cones(sets), O.Settings(kkt_solver="cg", **st))
    return res, ref


def test_unconstrained_qp_no_rows():
    rng = np.random.default_rng(0)
    n = 7
    S = rng.standard_normal((n, n)); P = S @ S.T + np.eye(n); q = rng.standard_normal(n)
    res, ref = _both(sp.csc_matrix(P), q, sp.csc_matrix((0, n)), np.zeros(0), [])
    assert res.status == ref.status == "Solved"
    assert np.linalg.norm(res.x + np.linalg.solve(P, q)) <= 1e-3
    assert np.linalg.norm(res.x - ref.x) <= 1e-6


def test_zero_dimensional_cones_and_single_variable():
    # min x^2 - 2x  s.t. x <= 0.5 (Nonnegatives on 0.5 - x), with empty sets of every simple kind in between
    sets = [cj.ZeroSet(0), cj.Nonnegatives(1), cj.SecondOrderCone(0), cj.Nonnegatives(0)]
    res, ref = _both(sp.csc_matrix([[2.0]]), np.array([-2.0]), sp.csc_matrix([[1.0]]), np.array([0.5]), sets)
    assert res.status == ref.status == "Solved"
    assert abs(res.x[0] - 0.5) < 1e-3 and abs(res.x[0] - ref.x[0]) < 1e-6
    assert abs(res.iter - ref.iter) <= 25


def test_lp_without_quadratic_term_and_dense_long_rows():
    # rows with more nonzeros than the 2048-entry LDS tile go through the chunked SpMV path inside the loop
    rng = np.random.default_rng(2)
    n, m = 3000, 12
    G = rng.standard_normal((m, n))                      # dense rows: 3000 nonzeros each
    x0 = rng.uniform(0.5, 1.5, n)
    A = sp.vstack([sp.csc_matrix(G), -sp.identity(n, format="csc")], format="csc")    # G x + s1 = b1 (s1 = 0) ; -x + s2 = 0 (s2 = x >= 0)
    b = np.concatenate([G @ x0, np.zeros(n)])
    q = rng.uniform(0.1, 1.0, n)
    sets = [cj.ZeroSet(m), cj.Nonnegatives(n)]
    res, ref = _both(sp.csc_matrix((n, n)), q, A, b, sets, max_iter=3000)
    assert res.status == ref.status
    assert abs(res.iter - ref.iter) <= 25
    assert abs(res.obj_val - ref.obj_val) <= 1e-4 * (1 + abs(ref.obj_val))


def test_repeated_optimize_and_warm_start_reuse_device_state():
    rng = np.random.default_rng(3)
    prob = util.random_qp(rng, 30, 2, 20, 20)
    model = cj.Model(); model.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], cj.Settings())
    r1 = cj.optimize(model)
    r2 = cj.optimize(model)                               # second call: warm start from the solution, KKT counters persist
    ws = O.Workspace(prob["P"], prob["q"], prob["A"], prob["b"], util.oracle_cones(prob["sets"]), O.Settings(kkt_solver="cg"))
    o1 = ws.optimize(); o2 = ws.optimize()
    assert r1.status == o1.status == "Solved" and r2.status == o2.status
    assert abs(r1.iter - o1.iter) <= 25 and abs(r2.iter - o2.iter) <= 25 and r2.iter <= r1.iter     # model_modifications.jl:30-33
    assert abs(r2.obj_val - o2.obj_val) <= 1e-4 * (1 + abs(o2.obj_val))
    # explicit warm start through the mirrored API
    model2 = cj.Model(); model2.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], cj.Settings())
    cj.warm_start_primal(model2, r1.x); cj.warm_start_dual(model2, r1.y)
    r3 = cj.optimize(model2)
    assert r3.status == "Solved" and r3.iter <= r1.iter


def test_abi_error_paths():
    F = cj._ffi
    h = cj.Handle(0)
    with pytest.raises(cj.CosmoHipError) as e:
        h.set_cones([F.NONNEG], [3], None, None)                  # set_problem missing
    assert e.value.code == 1
    h.set_problem(sp.identity(2, format="csc"), np.zeros(2), sp.csc_matrix((3, 2)), np.zeros(3))
    with pytest.raises(cj.CosmoHipError):
        h.set_cones([F.NONNEG], [4], None, None)                  # dimensions do not sum to m
    with pytest.raises(cj.CosmoHipError) as e:
        h.set_cones([12], [3], None, None)                        # unknown cone type: outside the hot path
    assert e.value.code == 6
    h.set_cones([F.NONNEG], [3], None, None)
    p = h.default_params(); p.adaptive_rho_interval = 0; p.adaptive_rho_fraction = -1.0
    with pytest.raises(cj.CosmoHipError) as e:
        h.set_params(p)                                           # the automatic rho interval (solver.jl:244-256) is honoured since round 5 -- with sane arguments
    assert e.value.code == 1
    p.adaptive_rho_fraction = 0.4
    h.set_params(p)                                               # (tests/test_gpu_auto_rho_interval.py has the behaviour)
    h.set_params(h.default_params())
    with pytest.raises(cj.CosmoHipError):
        h.optimize()                                              # set_iterates missing
