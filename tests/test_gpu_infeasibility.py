"""GPU tests of the infeasibility certificates (src/infeasibility.jl, scheduled by src/solver.jl:145-148,326-349):
the reference's own Box goldens (test/UnitTests/qp-box.jl:35-106) through the mirrored model interface, and SOC / PSD
instances against the oracle (same status, same detection iteration)."""
import numpy as np
import pytest
import scipy.sparse as sp

import cosmo_jl_amd as cj
from oracle import cosmo_oracle as O
from tests import util

pytestmark = pytest.mark.gpu


def _box_model(Am, b, P, q, l, u, **st):
    model = cj.Model()
    cj.assemble(model, P, np.array(q, dtype=float), cj.Constraint(sp.csc_matrix(np.array(Am, dtype=float)), b, cj.Box(l, u)), settings=cj.Settings(**st))
    return model


def test_box_primal_infeasible_goldens():
    r = cj.optimize(_box_model([[1.0, 0], [1, 0]], [2.0, 0], np.eye(2), [1.0, -1], [0.0, 0], [1.0, 1]))
    assert r.status == "Primal_infeasible" and r.obj_val == np.inf                      # qp-box.jl:50
    r = cj.optimize(_box_model([[1.0, 0], [1, 0]], [0.0, 0], np.eye(2), [1.0, -1], [0.0, 2], [1.0, 3]))
    assert r.status == "Primal_infeasible"                                              # qp-box.jl:68


@pytest.mark.parametrize("st", [dict(check_infeasibility=20, scaling=0), dict(check_infeasibility=40, scaling=10)])
def test_box_dual_infeasible_goldens(st):
    r = cj.optimize(_box_model(np.eye(2), [1.0, 1], np.zeros((2, 2)), [1.0, 1], [0.0, -np.inf], [1.0, 3], **st))
    assert r.status == "Dual_infeasible" and r.obj_val == -np.inf                       # qp-box.jl:87,105


def _vs_oracle(P, q, cons_model, cons_oracle, **st):
    model = cj.Model(); cj.assemble(model, P, q, cons_model, settings=cj.Settings(**st))
    res = cj.optimize(model)
    A, b, cones = O.assemble(cons_oracle)
    ref = O.solve(P, q, A, b, cones, O.Settings(kkt_solver="cg", **st))
    assert res.status == ref.status, (res.status, ref.status)
    assert res.iter == ref.iter
    return res, ref


def test_soc_and_psd_infeasible_problems_match_oracle():
    # primal infeasible SOCP: (t, v) in SOC(3) with t fixed to -1
    n = 3
    A1 = np.eye(3); b1 = np.zeros(3)                       # x in SOC
    A2 = np.array([[1.0, 0, 0]]); b2 = np.array([1.0])     # x1 + 1 = 0
    res, ref = _vs_oracle(np.zeros((n, n)), np.array([0.0, 1.0, 1.0]),
                          [cj.Constraint(A1, b1, cj.SecondOrderCone), cj.Constraint(A2, b2, cj.ZeroSet)],
                          [O.Constraint(A1, b1, O.SecondOrderCone(3)), O.Constraint(A2, b2, O.ZeroSet(1))])
    assert ref.status == "Primal_infeasible"
    # dual infeasible SOCP: minimise -t over the cone (unbounded)
    res, ref = _vs_oracle(np.zeros((n, n)), np.array([-1.0, 0.0, 0.0]),
                          [cj.Constraint(A1, b1, cj.SecondOrderCone)], [O.Constraint(A1, b1, O.SecondOrderCone(3))])
    assert ref.status == "Dual_infeasible"
    # primal infeasible SDP (3x3, svec variables): X psd and X11 = -1
    nt = 6
    Apsd = np.eye(nt); bpsd = np.zeros(nt)
    Aeq = np.zeros((1, nt)); Aeq[0, 0] = 1.0; beq = np.array([1.0])
    res, ref = _vs_oracle(np.zeros((nt, nt)), np.ones(nt),
                          [cj.Constraint(Apsd, bpsd, cj.PsdConeTriangle), cj.Constraint(Aeq, beq, cj.ZeroSet)],
                          [O.Constraint(Apsd, bpsd, O.PsdConeTriangle(nt)), O.Constraint(Aeq, beq, O.ZeroSet(1))])
    assert ref.status == "Primal_infeasible"
    # dual infeasible SDP with a 20x20 cone (workgroup eigen path): minimise -trace(X) over X psd
    d = 20; nt = d * (d + 1) // 2
    q = -cj.problems.svec(np.eye(d))
    res, ref = _vs_oracle(sp.csc_matrix((nt, nt)), q, [cj.Constraint(sp.identity(nt, format="csc"), np.zeros(nt), cj.PsdConeTriangle)],
                          [O.Constraint(sp.identity(nt, format="csc"), np.zeros(nt), O.PsdConeTriangle(nt))])
    assert ref.status == "Dual_infeasible"


def test_feasible_problems_are_not_flagged():
    rng = np.random.default_rng(1)
    prob = util.random_qp(rng, 40, 3, 30, 30, soc_dims=(4, 6), psd_tri_dims=(5,))
    model = cj.Model(); model.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], cj.Settings(check_infeasibility=10))
    res = cj.optimize(model)
    ref = O.solve(prob["P"], prob["q"], prob["A"], prob["b"], util.oracle_cones(prob["sets"]), O.Settings(kkt_solver="cg", check_infeasibility=10))
    assert res.status == ref.status == "Solved" and abs(res.iter - ref.iter) <= 25


# ---- the reference's randomised infeasible-by-construction families (InfeasibilityTests/*.jl; our RNG, tests/infeasible_instances.py)
from tests import infeasible_instances as INF   # noqa: E402

_SETS = {INF.ZERO: cj.ZeroSet, INF.NONNEG: cj.Nonnegatives, INF.SOC: cj.SecondOrderCone, INF.PSD_SQUARE: cj.PsdCone}


@pytest.mark.parametrize("family,seed", INF.CASES)
def test_infeasible_families_match_oracle(family, seed):
    gen, accepted, _ = INF.FAMILIES[family]
    P, q, cons = gen(seed)
    st = dict(max_iter=2000, eps_abs=1e-5, eps_rel=1e-5)
    tight = dict(tol_constant=1e-10, tol_exponent=0.0)
    model = cj.Model()
    cj.assemble(model, P, q, [cj.Constraint(A, b, _SETS[k]) for (A, b, k, d) in cons],
                settings=cj.Settings(kkt_solver=cj.with_options(cj.CGIndirectKKTSolver, **tight), **st))
    res = cj.optimize(model)
    A, b, cones = O.assemble([O.Constraint(A, b, O.Cone(k, d, constr_type=(np.zeros(d, dtype=bool) if k == O.NONNEG else None)))
                              for (A, b, k, d) in cons])
    ref = O.solve(P, q, A, b, cones, O.Settings(kkt_solver="cg", **tight, **st))
    assert res.status == ref.status and res.status in accepted, (res.status, ref.status)
    # the iterates of an infeasible problem diverge (|w| ~ 1e10 and growing), which amplifies rounding differences: allow the
    # certificate to fire one check_infeasibility interval apart
    assert abs(res.iter - ref.iter) <= 40, (res.iter, ref.iter)
