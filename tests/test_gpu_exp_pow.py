"""GPU tests of the exponential / power cones and their duals (src/convexset.jl:497-779): projection parity
against the oracle through cosmo_hip_project, the reference's own problem goldens (test/UnitTests/exp_cone.jl,
pow_cone.jl) through the mirrored model interface, and status / iteration parity of those problems with the oracle."""
import math

import numpy as np
import pytest
import scipy.sparse as sp

import cosmo_jl_amd as cj
from oracle import cosmo_oracle as O

pytestmark = pytest.mark.gpu
F = cj._ffi


def _project_many(kinds, alphas, X):
    """X: (ncones, 3).  One handle whose composite set is the listed 3-d cones."""
    nc = len(kinds)
    m = 3 * nc
    h = cj.Handle(0)
    A = sp.identity(m, format="csc")
    h.set_problem(sp.csc_matrix((m, m)), np.zeros(m), A, np.zeros(m))
    h.set_cones(kinds, [3] * nc, cone_param=alphas)
    s, _rank, case = h.project(X.reshape(-1).copy())
    h.close()
    return s.reshape(nc, 3), np.asarray(case)


def _oracle_project(kinds, alphas, X):
    out = X.copy()
    cases = []
    for k, a, row in zip(kinds, alphas, out):
        cone = O.Cone(k, 3, alpha=a, max_iter=100 if k in (O.EXP, O.DUAL_EXP) else 20, tol=1e-8)
        O.project_cone(row, cone)
    return out


@pytest.mark.parametrize("kind", [F.EXP, F.DUAL_EXP, F.POW, F.DUAL_POW])
def test_projection_matches_oracle(kind):
    rng = np.random.default_rng(10 + kind)
    nc = 3000
    X = -25 + 50 * rng.random((nc, 3))                       # the sampling of test/UnitTests/sets.jl:89,101
    X[:50] *= 1e-3                                           # small vectors
    X[50:60, 2] = 0.0                                        # z == 0 shortcut of the power cone (:642-646)
    X[60:70, 1] = 0.0                                        # y == 0 boundary of K_exp
    alphas = (0.1 + 0.85 * rng.random(nc)) if kind in (F.POW, F.DUAL_POW) else np.zeros(nc)
    got, case = _project_many([kind] * nc, alphas, X)
    ref = _oracle_project([kind] * nc, alphas, X)
    assert set(np.unique(case)) <= {1, 2, 3, 4} and len(np.unique(case)) >= 3
    # same algorithm, device libm instead of openlibm: the iterations stop at 1e-8, and exp(x/y) amplifies that for y -> 0
    scale = np.maximum(1.0, np.abs(X).max(axis=1, keepdims=True))
    err = np.abs(got - ref) / scale
    assert np.quantile(err, 0.99) < 1e-9
    assert err.max() < 1e-5
    # the results lie in the cone, up to the handful of y -> 0 exponential-cone draws where the reference algorithm itself
    # (bisection to 1e-8 on lambda) leaves a larger violation -- the device must not be worse than the oracle there
    cone = O.Cone(kind, 3, alpha=0.0)
    bad = bad_ref = 0
    for row, rrow, a in zip(got, ref, alphas):
        cone.alpha = a
        bad += not O.in_cone(row, cone, 1e-3)
        bad_ref += not O.in_cone(rrow, cone, 1e-3)
    assert bad <= bad_ref + 1 and bad <= nc // 500


def test_mixed_composite_set_with_other_cones():
    # 3-d cones interleaved with SOC / Nonnegatives / PSD slices in one composite set: offsets must line up
    rng = np.random.default_rng(5)
    kinds = [F.NONNEG, F.EXP, F.SOC, F.POW, F.PSD_TRIANGLE, F.DUAL_EXP, F.DUAL_POW]
    dims = [4, 3, 5, 3, 6, 3, 3]
    alphas = [0, 0, 0, 0.3, 0, 0, 0.7]
    m = sum(dims)
    x = rng.normal(size=m) * 3
    h = cj.Handle(0)
    h.set_problem(sp.csc_matrix((m, m)), np.zeros(m), sp.identity(m, format="csc"), np.zeros(m))
    h.set_cones(kinds, dims, cone_param=alphas)
    s, rank, case = h.project(x.copy())
    h.close()
    ref = x.copy()
    cones = [O.Cone(k, d, alpha=a, max_iter=100 if k in (O.EXP, O.DUAL_EXP) else 20, tol=1e-8,
                    constr_type=(np.zeros(d, dtype=bool) if k == O.NONNEG else None)) for k, d, a in zip(kinds, dims, alphas)]
    O.project(ref, cones)
    assert np.allclose(s, ref, atol=1e-9)
    assert case[0] == -1 and case[4] == -1 and all(case[i] in (1, 2, 3, 4) for i in (1, 3, 5, 6))


def test_power_cone_rejects_bad_alpha():
    h = cj.Handle(0)
    h.set_problem(sp.csc_matrix((3, 3)), np.zeros(3), sp.identity(3, format="csc"), np.zeros(3))
    with pytest.raises(cj.CosmoHipError):
        h.set_cones([F.POW], [3], cone_param=[1.0])
    with pytest.raises(cj.CosmoHipError):
        h.set_cones([F.POW], [3])                            # no alpha at all
    with pytest.raises(cj.CosmoHipError):
        h.set_cones([F.EXP], [4])
    h.close()
    with pytest.raises(ValueError):
        cj.PowerCone(0.0)


def _solve_both(P, q, cons_model, cons_oracle, **st):
    model = cj.Model(); cj.assemble(model, P, q, cons_model, settings=cj.Settings(**st))
    res = cj.optimize(model)
    A, b, cones = O.assemble(cons_oracle)
    ref = O.solve(P, q, A, b, cones, O.Settings(kkt_solver="cg", **st))
    assert res.status == ref.status, (res.status, ref.status)
    assert abs(res.iter - ref.iter) <= 25, (res.iter, ref.iter)       # one check_termination interval
    return res, ref


E3 = sp.identity(3, format="csc")
P0 = sp.csc_matrix((3, 3))


def test_exp_cone_goldens():
    # exp_cone.jl:19-42  obj -5 (atol 1e-2)
    A2 = sp.csc_matrix(np.array([[0, 1.0, 0], [0, 0, 1]])); b2 = np.array([-1.0, -math.exp(5)])
    res, ref = _solve_both(P0, np.array([-1.0, 0, 0]),
                           [cj.Constraint(E3, np.zeros(3), cj.ExponentialCone), cj.Constraint(A2, b2, cj.ZeroSet)],
                           [O.Constraint(E3, np.zeros(3), O.ExponentialCone()), O.Constraint(A2, b2, O.ZeroSet(2))],
                           eps_abs=1e-4, eps_rel=1e-4)
    assert res.status == "Solved" and abs(res.obj_val + 5.0) < 1e-2
    assert abs(res.obj_val - ref.obj_val) < 1e-6 * max(1.0, abs(ref.obj_val)) or res.iter != ref.iter
    # exp_cone.jl:47-76
    res, _ = _solve_both(P0, np.array([1.0, 0, 0]),
                         [cj.Constraint(E3, np.zeros(3), cj.ExponentialCone), cj.Constraint(np.array([[0, -1.0, 0]]), [-1.0], cj.ZeroSet),
                          cj.Constraint(np.array([[0, 0, -1.0]]), [1.0], cj.ZeroSet)],
                         [O.Constraint(E3, np.zeros(3), O.ExponentialCone()), O.Constraint(np.array([[0, -1.0, 0]]), [-1.0], O.ZeroSet(1)),
                          O.Constraint(np.array([[0, 0, -1.0]]), [1.0], O.ZeroSet(1))])
    assert res.status == "Primal_infeasible"
    # exp_cone.jl:78-104
    res, _ = _solve_both(P0, np.array([1.0, 0, 0]),
                         [cj.Constraint(E3, [0, 0, -0.2], cj.ExponentialCone), cj.Constraint(-E3, [0, 0, -0.3], cj.ExponentialCone)],
                         [O.Constraint(E3, [0, 0, -0.2], O.ExponentialCone()), O.Constraint(-E3, [0, 0, -0.3], O.ExponentialCone())])
    assert res.status == "Primal_infeasible"
    # exp_cone.jl:106-124
    res, _ = _solve_both(P0, np.array([0, 0, -1.0]), [cj.Constraint(E3, np.zeros(3), cj.ExponentialCone)],
                         [O.Constraint(E3, np.zeros(3), O.ExponentialCone())])
    assert res.status == "Dual_infeasible"


def test_dual_exp_cone_goldens():
    A2 = sp.csc_matrix(np.array([[1.0, 0, 0], [0, 0, 1]])); b2 = np.array([1.0, -math.exp(5)])
    res, _ = _solve_both(P0, np.array([0, 1.0, 0]),
                         [cj.Constraint(E3, np.zeros(3), cj.DualExponentialCone), cj.Constraint(A2, b2, cj.ZeroSet)],
                         [O.Constraint(E3, np.zeros(3), O.DualExponentialCone()), O.Constraint(A2, b2, O.ZeroSet(2))])
    assert res.status == "Solved" and abs(res.obj_val + 6.0) < 1e-3                     # exp_cone.jl:154-155
    A2 = sp.csc_matrix(np.array([[1.0, 0, 0], [0, 1, 0]])); b2 = np.array([-1.0, -2.0])
    res, _ = _solve_both(P0, np.ones(3),
                         [cj.Constraint(E3, np.zeros(3), cj.DualExponentialCone), cj.Constraint(A2, b2, cj.ZeroSet)],
                         [O.Constraint(E3, np.zeros(3), O.DualExponentialCone()), O.Constraint(A2, b2, O.ZeroSet(2))])
    assert res.status == "Primal_infeasible"                                            # exp_cone.jl:184


def test_power_cone_goldens():
    n = 6
    A1 = sp.csc_matrix((np.ones(3), (np.arange(3), np.arange(3))), shape=(3, n))
    A2 = sp.csc_matrix((np.ones(3), (np.arange(3), np.arange(3, 6))), shape=(3, n))
    a3 = np.array([[1.0, 2, 0, 3, 0, 0]]); a4 = np.array([[0, 0, 0, 0, 1.0, 0]])
    q = np.zeros(n); q[2] = q[5] = -1.0
    res, _ = _solve_both(sp.csc_matrix((n, n)), q,
                         [cj.Constraint(A1, np.zeros(3), cj.PowerCone(0.6)), cj.Constraint(A2, np.zeros(3), cj.PowerCone(0.1)),
                          cj.Constraint(a3, [-3.0], cj.ZeroSet), cj.Constraint(a4, [-1.0], cj.ZeroSet)],
                         [O.Constraint(A1, np.zeros(3), O.PowerCone(0.6)), O.Constraint(A2, np.zeros(3), O.PowerCone(0.1)),
                          O.Constraint(a3, [-3.0], O.ZeroSet(1)), O.Constraint(a4, [-1.0], O.ZeroSet(1))], max_iter=5000)
    assert res.status == "Solved" and abs(res.obj_val + 1.8458) < 1e-3                  # pow_cone.jl:53-54
    res, _ = _solve_both(P0, np.array([0, 0, -1.0]),
                         [cj.Constraint(E3, np.zeros(3), cj.PowerCone(0.8)), cj.Constraint(E3, [-1.0, -1, -2], cj.ZeroSet)],
                         [O.Constraint(E3, np.zeros(3), O.PowerCone(0.8)), O.Constraint(E3, [-1.0, -1, -2], O.ZeroSet(3))])
    assert res.status == "Primal_infeasible"                                            # pow_cone.jl:94
    res, _ = _solve_both(P0, np.array([0, 0, 1.0]), [cj.Constraint(E3, np.zeros(3), cj.PowerCone(0.8))],
                         [O.Constraint(E3, np.zeros(3), O.PowerCone(0.8))])
    assert res.status == "Dual_infeasible"                                              # pow_cone.jl:111
    A2 = np.array([[1.0, 0, 0], [0, 1, 0]])
    res, _ = _solve_both(P0, np.array([0, 0, -1.0]),
                         [cj.Constraint(E3, np.zeros(3), cj.DualPowerCone(0.8)), cj.Constraint(A2, [-0.8, -0.2], cj.ZeroSet)],
                         [O.Constraint(E3, np.zeros(3), O.DualPowerCone(0.8)), O.Constraint(A2, [-0.8, -0.2], O.ZeroSet(2))])
    assert res.status == "Solved" and abs(res.obj_val + 1.0) < 1e-3                     # pow_cone.jl:136-137


# ---------------------------------------------------------------------------------------------------------------------
# batch mode (round 4): the 3-dimensional cones inside the persistent batch kernels (csrc/cone3.h called from csrc/batch.hip), projection AND
# certificates -- the reference's goldens above, every one as a batch of three copies (the third with a scaled objective: its own rho / status)
# ---------------------------------------------------------------------------------------------------------------------
def _batch_golden(P, q, cons, want, obj=None, atol=1e-2, **st):
    mods = []
    for scale in (1.0, 1.0, 2.0):
        md = cj.Model(); cj.assemble(md, P * scale, np.asarray(q, dtype=float) * scale, cons(), settings=cj.Settings(**st)); mods.append(md)
    res = cj.optimize_batch(mods)
    one = cj.Model(); cj.assemble(one, P, np.asarray(q, dtype=float), cons(), settings=cj.Settings(**st))
    r1 = cj.optimize(one)
    for k, r in enumerate(res):
        assert r.status == want == r1.status, (k, r.status, r1.status)
        if obj is not None:
            assert abs(r.obj_val / (2.0 if k == 2 else 1.0) - obj) < (atol if k < 2 else 5 * atol), (k, r.obj_val)     # (the scaled copy stops on its own trajectory)
    assert abs(res[0].iter - r1.iter) <= 25 and res[0].iter == res[1].iter
    if obj is not None:
        assert np.linalg.norm(res[0].x - r1.x) <= 1e-3 * max(1.0, np.linalg.norm(r1.x))
    return res


@pytest.mark.parametrize("env", [dict(), dict(COSMO_HIP_BATCH_REG="0"), dict(COSMO_HIP_BATCH_LDS="0")])
def test_batch_exp_and_power_cone_goldens(env, monkeypatch):
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    A2 = sp.csc_matrix(np.array([[0, 1.0, 0], [0, 0, 1]])); b2 = np.array([-1.0, -math.exp(5)])
    _batch_golden(P0, [-1.0, 0, 0], lambda: [cj.Constraint(E3, np.zeros(3), cj.ExponentialCone), cj.Constraint(A2, b2, cj.ZeroSet)], "Solved", -5.0,
                  eps_abs=1e-4, eps_rel=1e-4)                                                                           # exp_cone.jl:19-42
    _batch_golden(P0, [1.0, 0, 0], lambda: [cj.Constraint(E3, np.zeros(3), cj.ExponentialCone), cj.Constraint(np.array([[0, -1.0, 0]]), [-1.0], cj.ZeroSet),
                                           cj.Constraint(np.array([[0, 0, -1.0]]), [1.0], cj.ZeroSet)], "Primal_infeasible")   # exp_cone.jl:47-76
    _batch_golden(P0, [0, 0, -1.0], lambda: [cj.Constraint(E3, np.zeros(3), cj.ExponentialCone)], "Dual_infeasible")           # exp_cone.jl:106-124
    A3 = sp.csc_matrix(np.array([[1.0, 0, 0], [0, 0, 1]])); b3 = np.array([1.0, -math.exp(5)])
    _batch_golden(P0, [0, 1.0, 0], lambda: [cj.Constraint(E3, np.zeros(3), cj.DualExponentialCone), cj.Constraint(A3, b3, cj.ZeroSet)], "Solved", -6.0,
                  atol=1e-3)                                                                                            # exp_cone.jl:154-155
    n = 6
    A1 = sp.csc_matrix((np.ones(3), (np.arange(3), np.arange(3))), shape=(3, n))
    A4 = sp.csc_matrix((np.ones(3), (np.arange(3), np.arange(3, 6))), shape=(3, n))
    a3 = np.array([[1.0, 2, 0, 3, 0, 0]]); a4 = np.array([[0, 0, 0, 0, 1.0, 0]])
    q = np.zeros(n); q[2] = q[5] = -1.0
    _batch_golden(sp.csc_matrix((n, n)), q, lambda: [cj.Constraint(A1, np.zeros(3), cj.PowerCone(0.6)), cj.Constraint(A4, np.zeros(3), cj.PowerCone(0.1)),
                                                    cj.Constraint(a3, [-3.0], cj.ZeroSet), cj.Constraint(a4, [-1.0], cj.ZeroSet)], "Solved", -1.8458,
                  atol=1e-3, max_iter=5000)                                                                             # pow_cone.jl:53-54
    _batch_golden(P0, [0, 0, -1.0], lambda: [cj.Constraint(E3, np.zeros(3), cj.PowerCone(0.8)), cj.Constraint(E3, [-1.0, -1, -2], cj.ZeroSet)],
                  "Primal_infeasible")                                                                                  # pow_cone.jl:94
    _batch_golden(P0, [0, 0, 1.0], lambda: [cj.Constraint(E3, np.zeros(3), cj.PowerCone(0.8))], "Dual_infeasible")        # pow_cone.jl:111


def test_batch_power_cone_api():
    B = F.Batch(1, 3, 3, 0)
    with pytest.raises(F.CosmoHipError) as e:
        B.set_cones([F.POW], [3], cone_param=[1.5])
    assert e.value.code == 1
    with pytest.raises(F.CosmoHipError) as e:
        B.set_cones([F.EXP], [4])
    assert e.value.code in (1, 6)
    B.close()
