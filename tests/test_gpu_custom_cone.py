"""User-defined cones through the AbstractConvexCone plugin surface (src/projections.jl:4-5): the reference's worked example
docs/src/literate/custom_cone.jl (Nonpositives) on the device library, against its documented answers and the oracle."""
import numpy as np
import pytest
import scipy.sparse as sp

import cosmo_jl_amd as cj
from oracle import cosmo_oracle as O
from tests import util

pytestmark = pytest.mark.gpu


class Nonpositives(cj.AbstractConvexCone):                 # custom_cone.jl:9-17
    def project(self, x):
        np.minimum(x, 0.0, out=x)


class NonpositivesFull(Nonpositives):                      # custom_cone.jl:62-68
    def in_dual(self, x, tol):
        return not np.any(x > -tol)

    def in_pol_recc(self, x, tol):
        return not np.any(x < tol)


def _o_nonpos(dim, full=True):
    def project(x):
        np.minimum(x, 0.0, out=x)
    if not full:
        return O.CustomCone(dim, project)
    return O.CustomCone(dim, project, in_dual=lambda x, tol: not np.any(x > -tol), in_pol_recc=lambda x, tol: not np.any(x < tol))


TIGHT = dict(tol_constant=1e-10, tol_exponent=0.0)


def _solve_both(P, q, cons_model, cons_oracle, **st):
    model = cj.Model()
    cj.assemble(model, P, q, cons_model, settings=cj.Settings(kkt_solver=cj.with_options(cj.CGIndirectKKTSolver, **TIGHT), **st))
    res = cj.optimize(model)
    A, b, cones = O.assemble(cons_oracle)
    ref = O.solve(P, q, A, b, cones, O.Settings(kkt_solver="cg", **TIGHT, **st))
    return res, ref


def test_custom_cone_lp_golden():
    # custom_cone.jl:19-49 -> x = (3, 2, 2), objective 7
    A1 = np.array([[1.0, 0, 0], [0, 1.0, 0]]); b1 = np.array([-3.0, -2.0])
    A2 = np.array([[1.0, 0, 1.0]]); b2 = np.array([-5.0])
    res, ref = _solve_both(sp.csc_matrix((3, 3)), -np.ones(3),
                           [cj.Constraint(A1, b1, Nonpositives), cj.Constraint(A2, b2, cj.ZeroSet)],
                           [O.Constraint(A1, b1, _o_nonpos(2, False)), O.Constraint(A2, b2, O.ZeroSet(1))])
    assert res.status == ref.status == "Solved"
    assert res.iter == ref.iter
    np.testing.assert_allclose(res.x, [3.0, 2.0, 2.0], atol=1e-3)
    np.testing.assert_allclose(res.x, ref.x, atol=1e-8)
    assert abs(-res.obj_val - 7.0) < 1e-3


def test_custom_cone_dual_infeasible_golden():
    # custom_cone.jl:70-89: min x s.t. x <= 3 -> :Dual_infeasible
    Ai = np.array([[1.0]]); bi = np.array([-3.0])
    res, ref = _solve_both(sp.csc_matrix((1, 1)), np.array([1.0]), [cj.Constraint(Ai, bi, NonpositivesFull)],
                           [O.Constraint(Ai, bi, _o_nonpos(1))])
    assert res.status == ref.status == "Dual_infeasible"
    assert res.iter == ref.iter


def test_custom_cone_without_membership_methods_never_certifies():
    # "If no further information about the new cone is provided, the infeasibility detection is disabled" (custom_cone.jl:52-54)
    Ai = np.array([[1.0]]); bi = np.array([-3.0])
    res, ref = _solve_both(sp.csc_matrix((1, 1)), np.array([1.0]), [cj.Constraint(Ai, bi, Nonpositives)],
                           [O.Constraint(Ai, bi, _o_nonpos(1, False))], max_iter=200)
    assert res.status == ref.status == "Max_iter_reached"


def test_custom_clone_of_nonnegatives_matches_builtin_on_device():
    # the same cone once as a user plugin (host callback every iteration) and once built in: identical trajectories
    class MyNonneg(cj.AbstractConvexCone):
        calls = 0

        def project(self, x):
            MyNonneg.calls += 1
            np.maximum(x, 0.0, out=x)

    rng = np.random.default_rng(3)
    n, m = 40, 70
    Am = sp.csc_matrix(rng.standard_normal((m, n))); x0 = rng.standard_normal(n)
    b = Am @ x0 + rng.uniform(0.1, 1.0, m)
    Pm = sp.identity(n, format="csc"); q = rng.standard_normal(n)
    out = []
    for K in (cj.Nonnegatives, MyNonneg):
        model = cj.Model()
        cj.assemble(model, Pm, q, [cj.Constraint(-Am, b, K)],
                    settings=cj.Settings(scaling=0, kkt_solver=cj.with_options(cj.CGIndirectKKTSolver, **TIGHT)))
        out.append(cj.optimize(model))
    r1, r2 = out
    assert r1.status == r2.status == "Solved" and r1.iter == r2.iter
    assert MyNonneg.calls >= r2.iter                       # one callback per iteration (+ safeguards / none here)
    np.testing.assert_allclose(r2.x, r1.x, rtol=0, atol=1e-13 * max(1.0, np.max(np.abs(r1.x))))


def test_custom_cone_in_a_mixed_scaled_problem_matches_oracle():
    # a user cone next to built-in cones, Ruiz scaling on the device (user cones get one scalar per cone)
    rng = np.random.default_rng(8)
    n = 30
    G1 = rng.standard_normal((10, n)); G2 = rng.standard_normal((6, n)); G3 = rng.standard_normal((8, n))
    x0 = rng.standard_normal(n)
    h1 = G1 @ x0 + rng.uniform(0.1, 1.0, 10)                 # G1 x - h1 <= 0        (Nonpositives)
    v = G2 @ x0; t = np.linalg.norm(v[1:]) + 0.5
    h2 = v.copy(); h2[0] -= t                                 # G2 x - h2 in SOC
    h3 = G3 @ x0                                              # G3 x == h3
    S = rng.standard_normal((n, n)); Pm = sp.csc_matrix(S @ S.T / n + 0.1 * np.eye(n)); q = rng.standard_normal(n)
    cm = [cj.Constraint(G1, -h1, NonpositivesFull), cj.Constraint(G2, -h2, cj.SecondOrderCone), cj.Constraint(G3, -h3, cj.ZeroSet)]
    co = [O.Constraint(G1, -h1, _o_nonpos(10)), O.Constraint(G2, -h2, O.SecondOrderCone(6)), O.Constraint(G3, -h3, O.ZeroSet(8))]
    res, ref = _solve_both(Pm, q, cm, co)
    assert res.status == ref.status == "Solved"
    assert res.iter == ref.iter
    np.testing.assert_allclose(res.x, ref.x, atol=1e-7 * max(1.0, np.max(np.abs(ref.x))))
    assert np.all(G1 @ res.x - h1 <= 1e-3)


def test_custom_cone_api_errors():
    h = cj.Handle(0)
    A = sp.identity(3, format="csc")
    h.set_problem(sp.identity(3, format="csc"), np.zeros(3), A, np.zeros(3))
    h.set_cones([cj._ffi.NONNEG, cj._ffi.CUSTOM], [1, 2])
    with pytest.raises(cj.CosmoHipError):                   # cone 0 is not a custom cone
        h.set_custom_cone(0, lambda x: None)
    p = h.default_params(); h.set_params(p)
    with pytest.raises(cj.CosmoHipError):                   # no projection callback installed yet
        h.project(np.ones(3))
    h.set_custom_cone(1, lambda x: np.minimum(x, 0.0, out=x))
    s = h.project(np.array([-1.0, 2.0, -3.0]))
    s = s[0] if isinstance(s, tuple) else s
    np.testing.assert_array_equal(s, [0.0, 0.0, -3.0])
