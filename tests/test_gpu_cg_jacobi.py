"""The OPT-IN Jacobi-preconditioned CG (kkt_kind COSMO_HIP_KKT_CG_JACOBI, csrc/cg_fold.hip + k_cg_upd<true>): IterativeSolvers' preconditioned
recurrence (PCGIterable) with Pl = Diagonal(diag(P + sigma I + A' rho A)) on the assembled reduced operator.  The reference calls cg! WITHOUT a
preconditioner (src/linear_solver/kktsolver_indirect.jl:70), so this is never the parity path; what IS held fixed -- and tested here -- is the
linear system, the warm start and the true-residual stopping rule ||r||_2 <= tol_k / ||rhs||:
  * solve! against a dense solve of the full KKT matrix in tight mode (test/UnitTests/kktsolver.jl:97-109);
  * tight-mode trajectories against the oracle's restated preconditioned recurrence (oracle.pcg_v09) at 1e-7, Krylov totals within the
    +-1-per-solve of a 1e-10 threshold, and against the literal unpreconditioned device path (both solve every system to 1e-10);
  * default schedule: same status, iteration count within one check interval, objective 1e-4, rho sequence -- with fewer Krylov iterations;
  * the preconditioner follows rho (device adaptation and the update_rho! entry point); no silent fallback where the operator is not assembled."""
import numpy as np
import pytest
import scipy.sparse as sp

import cosmo_jl_amd as cj
from oracle import cosmo_oracle as O
from tests import util
from tests.test_gpu_cg_fold import PROBLEMS

pytestmark = pytest.mark.gpu

TIGHT_PCG = dict(kkt_solver=cj.with_options(cj.CGJacobiKKTSolver, tol_constant=1e-10, tol_exponent=0.0))
TIGHT_CG = dict(kkt_solver=cj.with_options(cj.CGIndirectKKTSolver, tol_constant=1e-10, tol_exponent=0.0))


def _run(prob, iters, **st_kw):
    st = cj.Settings(max_iter=iters, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9, **st_kw)
    md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], st)
    return md, cj.optimize(md)


@pytest.mark.parametrize("name", sorted(PROBLEMS))
def test_jacobi_pcg_tight_trajectory_vs_oracle_and_vs_the_literal_cg(name):
    prob = PROBLEMS[name]()
    iters = 60
    md, r = _run(prob, iters, **TIGHT_PCG)
    assert md.handle.fold_stats()["enabled"] == 1
    ref = O.solve(prob["P"], prob["q"], prob["A"], prob["b"], util.oracle_cones(prob["sets"]),
                  O.Settings(max_iter=iters, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9, kkt_solver="cg_jacobi", tol_constant=1e-10, tol_exponent=0.0))
    assert r.iter == ref.iter == iters
    for a, b in ((r.x, ref.x), (r.s, ref.s), (r.y, ref.y)):
        assert np.max(np.abs(a - b)) <= 1e-7 * max(1.0, float(np.max(np.abs(b))))
    assert len(r.info.rho_updates) == len(ref.rho_updates)
    ktot = int(np.sum(ref.cg_iters))
    assert abs(r.kkt_iters_total - ktot) <= iters + 1 + 0.01 * ktot, (r.kkt_iters_total, ktot)
    # the literal recurrence on the same problem: another Krylov method for the same systems -- same trajectory at the tight tolerance
    md0, r0 = _run(prob, iters, **TIGHT_CG)
    for a, b in ((r.x, r0.x), (r.s, r0.s), (r.y, r0.y)):
        assert np.max(np.abs(a - b)) <= 1e-7 * max(1.0, float(np.max(np.abs(b))))
    print("%s: Krylov iterations in %d ADMM iterations: Jacobi-PCG %d (oracle %d), literal cg! %d" % (name, iters, r.kkt_iters_total, ktot, r0.kkt_iters_total))


def test_jacobi_pcg_kkt_solve_against_a_dense_solve_and_update_rho():
    prob = cj.problems.chordal_sdp(ncliques=6, dmin=6, dmax=20, sep_min=1, sep_max=2, n_total=600, n_zero=6, n_nonneg=12)
    md, _ = _run(prob, 3, scaling=0, adaptive_rho=False, **TIGHT_PCG)
    h = md.handle
    n, m = md.n, md.m
    rng = np.random.default_rng(5)
    for trial in range(2):
        rho = h.get_rho_vec() if trial == 0 else 10.0 ** rng.uniform(-3, 2, m)
        if trial == 1:
            h.update_rho(rho)                                  # AbstractKKTSolver.update_rho!: assembled values AND the preconditioner follow
        K = O.assemble_kkt_full(sp.csc_matrix(prob["P"]), sp.csc_matrix(prob["A"]), 1e-6, rho).toarray()
        rhs = rng.standard_normal(n + m)
        sol, its = h.kkt_solve(rhs)
        ref = np.linalg.solve(K, rhs)
        assert its > 0 and np.linalg.norm(sol - ref) <= 1e-7 * np.linalg.norm(ref)
        # against the oracle's preconditioned recurrence on the same system (same start: the previous solution)
    # iteration counts of single solves: device vs oracle.pcg_v09 from a zero start
    md2, _ = _run(prob, 0 + 1, scaling=0, adaptive_rho=False, **TIGHT_PCG)
    rho = md2.handle.get_rho_vec()
    ops = O.Operators(sp.csc_matrix(prob["P"]), sp.csc_matrix(prob["A"]))
    kk = O.IndirectReducedKKT(ops, n, m, 1e-6, rho, "CG_JACOBI", 1e-10, 0.0)
    kk.previous_solution[:] = md2.handle.get_kkt_solution()[:n]
    kk.iteration_counter = 3                                   # (irrelevant at tol_exponent = 0)
    rhs = rng.standard_normal(n + m)
    ref = kk.solve(rhs)
    sol, its = md2.handle.kkt_solve(rhs)
    assert abs(its - kk.last_iters) <= 2 + 0.02 * kk.last_iters, (its, kk.last_iters)     # a 1e-10 threshold on a ~240-iteration solve: a few iterations of rounding
    assert np.linalg.norm(sol - ref) <= 1e-8 * np.linalg.norm(ref) * 10


def test_jacobi_pcg_default_schedule_same_answer_fewer_krylov_iterations():
    prob = PROBLEMS["chordal_sdp"]()
    res = {}
    for kkt in (cj.CGJacobiKKTSolver, cj.CGIndirectKKTSolver):
        md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], cj.Settings(kkt_solver=kkt, max_iter=3000))
        res[kkt] = cj.optimize(md)
    rp, rc = res[cj.CGJacobiKKTSolver], res[cj.CGIndirectKKTSolver]
    ref = O.solve(prob["P"], prob["q"], prob["A"], prob["b"], util.oracle_cones(prob["sets"]), O.Settings(kkt_solver="cg_jacobi", max_iter=3000))
    assert rp.status == rc.status == ref.status == "Solved"
    assert abs(rp.iter - ref.iter) <= 25 and abs(rp.iter - rc.iter) <= 50       # vs its own oracle: one interval; vs the literal CG: another inexact solver
    assert abs(rp.obj_val - ref.obj_val) <= 1e-4 * (1 + abs(ref.obj_val)) and abs(rp.obj_val - rc.obj_val) <= 1e-3 * (1 + abs(rc.obj_val))
    assert len(rp.info.rho_updates) == len(ref.rho_updates) >= 2                # rho changed on the device: the preconditioner was rebuilt
    assert rp.kkt_iters_total / max(rp.iter, 1) < rc.kkt_iters_total / max(rc.iter, 1), (rp.kkt_iters_total, rc.kkt_iters_total)
    print("default schedule: Jacobi-PCG %d Krylov iterations in %d ADMM iterations, literal cg! %d in %d" % (rp.kkt_iters_total, rp.iter, rc.kkt_iters_total, rc.iter))


def test_jacobi_pcg_refuses_an_operator_that_is_not_assembled():
    """config-2-like rows: A' rho A is too dense to assemble -> UNSUPPORTED, never a silent switch to the unpreconditioned recurrence."""
    prob = cj.problems.sparse_box_qp(n=2000, m=4000, nnz=40000, seed=3)
    md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], cj.Settings(kkt_solver=cj.CGJacobiKKTSolver, max_iter=5))
    with pytest.raises(Exception, match="assembled"):
        cj.optimize(md)


def test_jacobi_pcg_graph_chain_equals_direct_launches(monkeypatch):
    """The captured chain of speculative Krylov iterations carries the preconditioned kernels too: bit-identical to direct launches."""
    prob = PROBLEMS["chordal_sdp"]()
    out = {}
    for graph in ("1", "0"):
        monkeypatch.setenv("COSMO_HIP_CG_GRAPH", graph)
        md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], cj.Settings(kkt_solver=cj.CGJacobiKKTSolver, max_iter=120, eps_abs=0, eps_rel=0))
        out[graph] = cj.optimize(md)
    a, b = out["1"], out["0"]
    assert a.kkt_iters_total == b.kkt_iters_total and np.array_equal(a.x, b.x) and np.array_equal(a.s, b.s)
