"""The opt-in single-reduction (Chronopoulos-Gear) CG (csrc/cg_sr.hip, kkt_kind COSMO_HIP_KKT_CG_SR) against the literal cg!
restatement: same operator, stopping rule (abstol = tol_k / ||rhs||, checked before each iteration) and warm start; algebraically
equal iterates, so it is held to the KKT tolerances of SURVEY 8c, not to bit equality (VERDICT r1, next-round item 4 iii)."""
import numpy as np
import pytest
import scipy.sparse as sp

import cosmo_jl_amd as cj
from oracle import cosmo_oracle as O
from tests import util

pytestmark = pytest.mark.gpu

PROBLEMS = {
    "box_qp": lambda: cj.problems.sparse_box_qp(n=3000, m=6000, nnz=50000, seed=9),
    "mixed_cones_qp": lambda: util.random_qp(np.random.default_rng(21), 400, 20, 150, 120, soc_dims=(7, 12, 30), density=0.03, p_shift=0.5),
    "chordal_sdp_split_operator": lambda: cj.problems.chordal_sdp(ncliques=12, dmin=4, dmax=70, sep_min=1, sep_max=3, n_total=2500, n_zero=40, n_nonneg=80),
}


@pytest.fixture(params=["assembled_operator", "split_operator"], autouse=True)
def operator_form(request, monkeypatch):
    """Both forms of the reduced operator under the single-reduction recurrence: M assembled as one matrix (csrc/cg_fold.hip: ONE launch
    per Krylov iteration, k_sr_M) where the problem qualifies, and the split / plain operator (two launches).  The literal solver it is
    compared with uses the same form."""
    monkeypatch.setenv("COSMO_HIP_OP_FOLD", "1" if request.param == "assembled_operator" else "0")
    return request.param


def _run(prob, kkt, iters, tight=True):
    kw = dict(tol_constant=1e-10, tol_exponent=0.0) if tight else {}
    st = cj.Settings(kkt_solver=cj.with_options(kkt, **kw), max_iter=iters, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9)
    md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], st)
    r = cj.optimize(md)
    return r, md


@pytest.mark.parametrize("name", sorted(PROBLEMS))
def test_kkt_solve_matches_the_dense_solve_and_the_literal_cg(name):
    prob = PROBLEMS[name]()
    r_sr, md_sr = _run(prob, cj.CGSingleReductionKKTSolver, 3)
    r_cg, md_cg = _run(prob, cj.CGIndirectKKTSolver, 3)
    rhs = np.random.default_rng(5).standard_normal(md_sr.n + md_sr.m)
    sol_sr, k_sr = md_sr.handle.kkt_solve(rhs)
    sol_cg, k_cg = md_cg.handle.kkt_solve(rhs)
    assert abs(k_sr - k_cg) <= 2 + 0.03 * k_cg, (k_sr, k_cg)                          # same stopping rule; at 1e-10 both sit on their rounding floors
    assert np.linalg.norm(sol_sr - sol_cg) <= 1e-8 * np.linalg.norm(sol_cg)
    if md_sr.n <= 3000:                                                                   # dense reference (test/UnitTests/kktsolver.jl:97-109 mirrors)
        n = md_sr.n
        h = md_sr.handle
        rho = h.get_rho_vec()
        # the scaled problem lives on the device: rebuild K from SpMV probes is expensive; compare residual of the reduced system instead
        x, nu = sol_sr[:n], sol_sr[n:]
        Ax = h.spmv(cj._ffi.MAT_A, x)
        assert np.linalg.norm(nu - rho * (Ax - rhs[n:])) <= 1e-9 * max(1.0, np.linalg.norm(nu))      # nu = rho .* (A x - rhs_s), kktsolver_indirect.jl:81-83
        Px = h.spmv(cj._ffi.MAT_P, x)
        lhs = Px + md_sr.settings.sigma * x + h.spmv(cj._ffi.MAT_AT, rho * Ax)
        red_rhs = rhs[:n] + h.spmv(cj._ffi.MAT_AT, rho * rhs[n:])
        assert np.linalg.norm(lhs - red_rhs) <= 1e-8 * np.linalg.norm(red_rhs)


@pytest.mark.parametrize("name", sorted(PROBLEMS))
def test_admm_trajectory_matches_the_literal_cg_in_tight_mode(name, operator_form):
    prob = PROBLEMS[name]()
    # 60 iterations: before these small problems converge to rounding level, where the adaptive-rho decision (a ratio of residuals
    # of size 1e-12) is noise for ANY two solvers and y = -rho .* (w - s) follows it (measured: x, s still agree to 1e-14 at 200)
    r_sr, md_sr = _run(prob, cj.CGSingleReductionKKTSolver, 60)
    r_cg, _ = _run(prob, cj.CGIndirectKKTSolver, 60)
    assert md_sr.handle.fold_stats()["enabled"] == (1 if (operator_form == "assembled_operator" and name == "chordal_sdp_split_operator") else 0)
    assert r_sr.iter == r_cg.iter == 60
    for a, b in ((r_sr.x, r_cg.x), (r_sr.s, r_cg.s), (r_sr.y, r_cg.y)):
        assert np.max(np.abs(a - b)) <= 1e-7 * max(1.0, float(np.max(np.abs(b))))                   # SURVEY 8c trajectory tolerance
    assert abs(r_sr.kkt_iters_total - r_cg.kkt_iters_total) <= 0.03 * r_cg.kkt_iters_total + 61     # +-1 per solve
    assert np.allclose(r_sr.info.rho_updates, r_cg.info.rho_updates, rtol=1e-6)


def test_default_tolerance_solve_matches_the_oracle():
    prob = cj.problems.sparse_box_qp(n=2000, m=4000, nnz=30000, seed=3)
    md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], cj.Settings(kkt_solver=cj.CGSingleReductionKKTSolver))
    r = cj.optimize(md)
    ref = O.solve(prob["P"], prob["q"], prob["A"], prob["b"], util.oracle_cones(prob["sets"]), O.Settings(kkt_solver="cg"))
    assert r.status == ref.status == "Solved" and abs(r.iter - ref.iter) <= 25
    assert abs(r.obj_val - ref.obj_val) <= 1e-4 * (1 + abs(ref.obj_val))


@pytest.mark.parametrize("chain_len", ["16", "3", "1"])
def test_one_launch_recurrence_captured_chain_is_bit_identical_to_direct_launches(chain_len, monkeypatch):
    """Round 6: on the assembled operator the single-reduction recurrence is ONE kernel per Krylov iteration (k_sr_M) and its speculative iterations go
    out as a captured chain with the iteration index read on the device (ctl->sr_k[parity]; records and partial slots alternate by parity, so an odd
    COSMO_HIP_CG_GRAPH_LEN is rounded up).  COSMO_HIP_CG_GRAPH=0 launches every kernel directly with the index as an argument: same bits, same
    Krylov counts, same number of enqueued iterations -- default schedule (rho adaptation) and tight mode."""
    monkeypatch.setenv("COSMO_HIP_OP_FOLD", "1")
    prob = PROBLEMS["chordal_sdp_split_operator"]()
    for kw in (dict(), dict(tol_constant=1e-10, tol_exponent=0.0)):
        res = {}
        for graph in ("0", "1"):
            monkeypatch.setenv("COSMO_HIP_CG_GRAPH", graph)
            monkeypatch.setenv("COSMO_HIP_CG_GRAPH_LEN", chain_len)
            st = cj.Settings(kkt_solver=cj.with_options(cj.CGSingleReductionKKTSolver, **kw), max_iter=90, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9)
            md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], st)
            r = cj.optimize(md)
            assert md.handle.fold_stats()["enabled"] == 1 and "one-launch single-reduction" in md.handle.kkt_recurrence() and "k_sr_M<" in md.handle.kkt_recurrence()
            res[graph] = (r, md.handle.get_stats())
        (r0, s0), (r1, s1) = res["0"], res["1"]
        assert r0.iter == r1.iter == 90
        assert np.array_equal(r0.x, r1.x) and np.array_equal(r0.s, r1.s) and np.array_equal(r0.y, r1.y)
        assert s0["kkt_iters_total"] == s1["kkt_iters_total"] > 0 and s0["kkt_budget_stalls"] == s1["kkt_budget_stalls"]
        assert list(r0.info.rho_updates) == list(r1.info.rho_updates)


def test_kkt_recurrence_names_the_kernels_that_run(monkeypatch):
    """cosmo_hip_kkt_recurrence: kkt_kind CG is the literal recurrence everywhere (two launches on an assembled operator, three on the split one); the
    one-launch form is opt-in (kkt_kind CG_SR) -- round 6 measured it as a default for assembled operators and rejected it."""
    monkeypatch.setenv("COSMO_HIP_OP_FOLD", "1")
    prob = PROBLEMS["chordal_sdp_split_operator"]()
    _, md = _run(prob, cj.CGIndirectKKTSolver, 2)
    assert md.handle.kkt_recurrence().startswith("cg: literal recurrence on the assembled operator, two launches per iteration, k_cg_dirM<")
    _, md = _run(PROBLEMS["box_qp"](), cj.CGIndirectKKTSolver, 2)
    assert md.handle.kkt_recurrence().startswith("cg: literal recurrence, three launches per iteration, k_cg_dirA + k_op_apply + k_cg_upd<false>")
    _, md = _run(prob, cj.CGJacobiKKTSolver, 2)
    assert "Jacobi" in md.handle.kkt_recurrence() and "k_cg_dirM<" in md.handle.kkt_recurrence()
