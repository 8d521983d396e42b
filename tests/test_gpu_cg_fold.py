"""The ASSEMBLED reduced operator of the CG solve (csrc/cg_fold.hip): M = P + diag(sigma + d) + Am' rho Am applied as ONE sparse
product (two launches per Krylov iteration) against the two dependent products of reduced_mul!
(src/linear_solver/kktsolver_indirect.jl:57-64) that the split / unsplit operators apply.  Same linear operator, different
association: the tests hold it to the KKT tolerances of SURVEY 8c -- dense solve in tight mode, Krylov iteration counts (+-1 per
solve), 1e-7 trajectories against the unfolded device path AND against the oracle, rho adaptation on the device included."""
import numpy as np
import pytest
import scipy.sparse as sp

import cosmo_jl_amd as cj
from oracle import cosmo_oracle as O
from tests import util

pytestmark = pytest.mark.gpu


def bounds_qp(n=1500, nrows=40, seed=3):
    """QP with a general sparse P, simple bounds on every variable (single-nonzero rows of A: the diagonal part of A' rho A) and a few
    dense-ish coupling rows (the matrix part Am)."""
    rng = np.random.default_rng(seed)
    S = sp.random(n, n, density=3.0 / n, random_state=rng, format="csc", data_rvs=lambda k: 0.2 * rng.standard_normal(k))
    Ps = (S + S.T).tocsc()
    P = (Ps + sp.diags(np.asarray(abs(Ps).sum(axis=1)).ravel() + rng.uniform(0.1, 1.0, n))).tocsc()
    P.sort_indices()
    G = sp.random(nrows, n, density=12.0 / n, random_state=rng, format="csc", data_rvs=rng.standard_normal)
    x0 = rng.standard_normal(n)
    A = sp.vstack([-sp.eye(n), -G], format="csc"); A.sort_indices()
    lo = np.concatenate([x0 - np.abs(rng.standard_normal(n)), G @ x0 - 0.1])
    hi = np.concatenate([x0 + np.abs(rng.standard_normal(n)), G @ x0 + 0.1])
    hi[n:n + nrows // 4] = lo[n:n + nrows // 4]                      # some equality rows: rho class x 1e3
    return dict(P=P, q=rng.standard_normal(n), A=A, b=np.zeros(n + nrows), sets=[cj.Box(lo, hi)])


PROBLEMS = {
    "chordal_sdp": lambda: cj.problems.chordal_sdp(ncliques=12, dmin=4, dmax=70, sep_min=1, sep_max=3, n_total=2500, n_zero=40, n_nonneg=80),
    "bounds_qp": bounds_qp,
}


def _run(monkeypatch, fold, prob, iters, **st_kw):
    monkeypatch.setenv("COSMO_HIP_OP_FOLD", fold)
    st = cj.Settings(max_iter=iters, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9, **st_kw)
    md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], st)
    r = cj.optimize(md)
    return md, r


TIGHT = dict(kkt_solver=cj.with_options(cj.CGIndirectKKTSolver, tol_constant=1e-10, tol_exponent=0.0))


@pytest.mark.parametrize("name", sorted(PROBLEMS))
def test_fold_matches_the_unfolded_operator_and_the_oracle(name, monkeypatch):
    prob = PROBLEMS[name]()
    iters = 60
    md1, r1 = _run(monkeypatch, "1", prob, iters, **TIGHT)
    fs = md1.handle.fold_stats()
    assert fs["enabled"] == 1 and fs["nnz"] >= md1.n, fs                 # the assembled operator really ran
    md0, r0 = _run(monkeypatch, "0", prob, iters, **TIGHT)
    assert md0.handle.fold_stats()["enabled"] == 0
    assert r1.iter == r0.iter == iters
    # Krylov work: +-1 iteration per solve at the 1e-10 stopping threshold
    assert abs(r1.kkt_iters_total - r0.kkt_iters_total) <= iters + 1, (r1.kkt_iters_total, r0.kkt_iters_total)
    for a, b in ((r1.x, r0.x), (r1.s, r0.s), (r1.y, r0.y)):
        assert np.max(np.abs(a - b)) <= 1e-7 * max(1.0, float(np.max(np.abs(b))))
    assert r1.info.rho_updates == pytest.approx(r0.info.rho_updates, rel=1e-6)
    # ... and the oracle (tight mode, ||dw|| <= 1e-7 ||w||)
    ref = O.solve(prob["P"], prob["q"], prob["A"], prob["b"], util.oracle_cones(prob["sets"]),
                  O.Settings(max_iter=iters, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9, kkt_solver="cg", tol_constant=1e-10, tol_exponent=0.0))
    assert ref.iter == r1.iter
    for a, b in ((r1.x, ref.x), (r1.s, ref.s), (r1.y, ref.y)):
        assert np.max(np.abs(a - b)) <= 1e-7 * max(1.0, float(np.max(np.abs(b))))
    assert len(r1.info.rho_updates) == len(ref.rho_updates)


@pytest.mark.parametrize("name", sorted(PROBLEMS))
def test_fold_kkt_solve_against_a_dense_solve(name, monkeypatch):
    """AbstractKKTSolver.solve! through the assembled operator vs numpy on the full KKT matrix (test/UnitTests/kktsolver.jl:97-109)."""
    prob = PROBLEMS[name]()
    if name == "chordal_sdp":
        prob = cj.problems.chordal_sdp(ncliques=6, dmin=6, dmax=20, sep_min=1, sep_max=2, n_total=600, n_zero=6, n_nonneg=12)
    md, _ = _run(monkeypatch, "1", prob, 5, scaling=0, adaptive_rho=False, **TIGHT)
    assert md.handle.fold_stats()["enabled"] == 1
    n, m = md.n, md.m
    if n + m > 4000:
        pytest.skip("dense reference too large")
    rho = md.handle.get_rho_vec()
    K = O.assemble_kkt_full(sp.csc_matrix(prob["P"]), sp.csc_matrix(prob["A"]), 1e-6, rho).toarray()
    rhs = np.random.default_rng(11).standard_normal(n + m)
    sol, its = md.handle.kkt_solve(rhs)
    ref = np.linalg.solve(K, rhs)
    assert its > 0
    assert np.linalg.norm(sol - ref) <= 1e-8 * np.linalg.norm(ref) * 10        # tight mode: ||dsol|| <= 1e-8 ||sol|| (x 10: dense-solve conditioning)


def test_fold_follows_rho_updates_and_default_schedule(monkeypatch):
    """Default settings (loose 1/k^1.5 Krylov tolerance, adaptive rho every 40 iterations decided on the device): the assembled
    values are rebuilt after every adaptation; status, iteration count and objective agree with the unfolded path and the oracle."""
    prob = PROBLEMS["chordal_sdp"]()
    out = {}
    for fold in ("1", "0"):
        monkeypatch.setenv("COSMO_HIP_OP_FOLD", fold)
        md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], cj.Settings(kkt_solver=cj.CGIndirectKKTSolver, max_iter=3000))
        out[fold] = (cj.optimize(md), md.handle.fold_stats()["enabled"])
    (r1, e1), (r0, e0) = out["1"], out["0"]
    assert e1 == 1 and e0 == 0
    ref = O.solve(prob["P"], prob["q"], prob["A"], prob["b"], util.oracle_cones(prob["sets"]), O.Settings(kkt_solver="cg", max_iter=3000))
    assert r1.status == r0.status == ref.status == "Solved"
    assert len(r1.info.rho_updates) == len(r0.info.rho_updates) == len(ref.rho_updates) >= 2      # rho really changed
    assert abs(r1.iter - r0.iter) <= 25 and abs(r1.iter - ref.iter) <= 25
    assert abs(r1.obj_val - ref.obj_val) <= 1e-4 * (1 + abs(ref.obj_val))


def test_fold_is_off_where_the_assembly_would_not_pay(monkeypatch):
    """config-2-like rows (10 nonzeros per row, no single-nonzero rows): neither split nor assembled."""
    monkeypatch.setenv("COSMO_HIP_OP_FOLD", "1")
    prob = cj.problems.sparse_box_qp(n=2000, m=4000, nnz=40000, seed=3)
    md, _ = _run(monkeypatch, "1", prob, 3)
    assert md.handle.fold_stats()["enabled"] == 0


def test_fold_follows_update_rho_through_the_plugin_entry_point(monkeypatch):
    """AbstractKKTSolver.update_rho! (src/linear_solver/kktsolver_indirect.jl:164-166 -> cosmo_hip_update_rho): the assembled values are
    rebuilt for an arbitrary rho vector handed over by the host, not only after the device's own adaptation."""
    monkeypatch.setenv("COSMO_HIP_OP_FOLD", "1")
    prob = cj.problems.chordal_sdp(ncliques=6, dmin=6, dmax=20, sep_min=1, sep_max=2, n_total=600, n_zero=6, n_nonneg=12)
    md, _ = _run(monkeypatch, "1", prob, 3, scaling=0, adaptive_rho=False, **TIGHT)
    h = md.handle
    assert h.fold_stats()["enabled"] == 1
    n, m = md.n, md.m
    rng = np.random.default_rng(31)
    rho = 10.0 ** rng.uniform(-3, 2, m)
    h.update_rho(rho)
    K = O.assemble_kkt_full(sp.csc_matrix(prob["P"]), sp.csc_matrix(prob["A"]), 1e-6, rho).toarray()
    rhs = rng.standard_normal(n + m)
    sol, its = h.kkt_solve(rhs)
    ref = np.linalg.solve(K, rhs)
    assert its > 0 and np.linalg.norm(sol - ref) <= 1e-7 * np.linalg.norm(ref)


@pytest.mark.parametrize("chain_len", ["16", "3", "1"])
@pytest.mark.parametrize("name", sorted(PROBLEMS))
def test_captured_chain_of_krylov_iterations_is_bit_identical_to_direct_launches(name, chain_len, monkeypatch):
    """The speculative Krylov iterations of a solve go out as a captured chain (hipGraph) with the iteration index read on the device
    (csrc/cg_fold.hip: fold_enqueue_iterations); COSMO_HIP_CG_GRAPH=0 launches every kernel directly with the index as an argument.
    Same kernels, same order, same number of enqueued iterations: iterates, Krylov counts, rho updates and stall counts are identical --
    default schedule (loose tolerance, rho adaptation every 40 iterations) and tight mode, chains of 16 (default), 3 and 1 iterations."""
    prob = PROBLEMS[name]()
    monkeypatch.setenv("COSMO_HIP_OP_FOLD", "1")
    for kw in (dict(), TIGHT):
        res = {}
        for graph in ("0", "1"):
            monkeypatch.setenv("COSMO_HIP_CG_GRAPH", graph)
            monkeypatch.setenv("COSMO_HIP_CG_GRAPH_LEN", chain_len)
            st = cj.Settings(max_iter=90, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9, **kw)
            md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], st)
            r = cj.optimize(md)
            assert md.handle.fold_stats()["enabled"] == 1
            res[graph] = (r, md.handle.get_stats())
        (r0, s0), (r1, s1) = res["0"], res["1"]
        assert r0.iter == r1.iter == 90
        assert np.array_equal(r0.x, r1.x) and np.array_equal(r0.s, r1.s) and np.array_equal(r0.y, r1.y)
        assert s0["kkt_iters_total"] == s1["kkt_iters_total"] > 0 and s0["kkt_budget_stalls"] == s1["kkt_budget_stalls"]
        assert s0["spmv_A"] == s1["spmv_A"]                                   # the same number of iterations was enqueued
        assert list(r0.info.rho_updates) == list(r1.info.rho_updates)


@pytest.mark.parametrize("name", sorted(PROBLEMS))
def test_krylov_budget_feedback_changes_the_number_of_launches_not_the_results(name, monkeypatch):
    """The loop enqueues a solve's Krylov iterations speculatively.  With the budget feedback (csrc/api.hip: solve_budget -- the count of
    every solve comes back through a pinned ring, the budget follows the counts of four solves ago) one long call enqueues far fewer no-op
    iterations than with the budget fixed per call (COSMO_HIP_BUDGET_FEEDBACK=0) and does not stall; iterates, Krylov counts and rho
    updates are the same bits either way, stalled or not."""
    prob = PROBLEMS[name]()
    res = {}
    for fb in ("0", "1"):
        monkeypatch.setenv("COSMO_HIP_BUDGET_FEEDBACK", fb)
        st = cj.Settings(max_iter=10 ** 6, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9, check_termination=10 ** 9)
        md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], st)
        cj.model.setup(md)
        h = md.handle
        h.set_iterates(md.x, md.s, md.mu); h.admm_init()
        h.admm_iterate_checked(5)
        s0 = h.get_stats()
        h.admm_iterate_checked(100)                                           # ONE call: rho checks at 40 and 80 inside it
        s1 = h.get_stats()
        w, _, s, mu = h.get_iterates()
        res[fb] = (w, s, mu, s1["kkt_iters_total"] - s0["kkt_iters_total"], s1["spmv_A"] - s0["spmv_A"], s1["kkt_budget_stalls"] - s0["kkt_budget_stalls"],
                   s1["rho_updates"])
        h.close()
    a, b = res["0"], res["1"]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert a[3] == b[3] > 0 and a[6] == b[6]                                    # Krylov iterations performed, rho updates
    assert b[5] <= 1                                                            # (next to) no stall with the feedback: a stall costs a window, never a result
    # enqueued <= 1.5 x performed + a few per solve; a stall (rare, trajectory dependent) re-enqueues the rest of the call once: twice that
    assert b[4] <= (1 + b[5]) * (1.5 * b[3] + 100 * 8)
    assert b[4] <= a[4]                                                         # and never more launches than the fixed budget


@pytest.mark.parametrize("kkt", ["literal", "jacobi"])
def test_padded_tile_major_operator_copy_is_bit_identical(kkt, monkeypatch):
    """Opt-in COSMO_HIP_FOLD_PAD=1: k_cg_dirM streams (col, val) from a tile-major padded copy whose addresses depend on the block index only (one
    dependent memory round trip less per workgroup; measured: no gain, hence opt-in); by default it reads the CSR arrays behind the tile descriptor.  Same arithmetic: iterates,
    Krylov counts and rho updates are the same bits -- default (rho-adapting) schedule, so the refresh of the padded values is exercised too."""
    prob = PROBLEMS["chordal_sdp"]()
    solver = cj.CGJacobiKKTSolver if kkt == "jacobi" else cj.CGIndirectKKTSolver
    out = {}
    for pad in ("1", "0"):
        monkeypatch.setenv("COSMO_HIP_FOLD_PAD", pad)
        md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], cj.Settings(kkt_solver=solver, max_iter=130, eps_abs=0.0, eps_rel=0.0))
        out[pad] = cj.optimize(md)
        assert md.handle.fold_stats()["enabled"] == 1
    a, b = out["1"], out["0"]
    assert a.kkt_iters_total == b.kkt_iters_total > 0 and list(a.info.rho_updates) == list(b.info.rho_updates) and len(a.info.rho_updates) >= 2
    assert np.array_equal(a.x, b.x) and np.array_equal(a.s, b.s) and np.array_equal(a.y, b.y)


@pytest.mark.parametrize("name", sorted(PROBLEMS))
def test_partially_assembled_operator_opt_in(name, monkeypatch):
    """Opt-in COSMO_HIP_FOLD_FACTOR=1 (round 6): rows of Am with >= 4 nonzeros stay FACTORED, M = Ms + Ad' rho Ad; their columns of the stored matrix gather
    {(Ad r), (Ad u_prev)} records, k_cg_updF forms (Ad r_new) as a fresh product from parity-buffered {r, u} records.  Same operator in another association:
    held to the fold tests' own tolerances against the fully assembled form, the unfolded operator and the oracle (tight mode: 1e-7 trajectories, Krylov totals
    +-1 per solve), and the captured chain (even length: records alternate by parity) against direct launches bit for bit."""
    prob = PROBLEMS[name]()
    iters = 60
    monkeypatch.setenv("COSMO_HIP_FOLD_FACTOR", "1")
    md1, r1 = _run(monkeypatch, "1", prob, iters, **TIGHT)
    fs = md1.handle.fold_stats()
    assert fs["enabled"] == 1 and fs["factored_rows"] > 0 and fs["stored_entries"] < fs["nnz"] and "partially assembled" in md1.handle.kkt_recurrence()
    monkeypatch.setenv("COSMO_HIP_FOLD_FACTOR", "0")
    md2, r2 = _run(monkeypatch, "1", prob, iters, **TIGHT)
    assert md2.handle.fold_stats()["factored_rows"] == 0 and md2.handle.fold_stats()["nnz"] == fs["nnz"]
    md0, r0 = _run(monkeypatch, "0", prob, iters, **TIGHT)
    for other in (r2, r0):
        assert abs(r1.kkt_iters_total - other.kkt_iters_total) <= iters + 1, (r1.kkt_iters_total, other.kkt_iters_total)
        for a, b in ((r1.x, other.x), (r1.s, other.s), (r1.y, other.y)):
            assert np.max(np.abs(a - b)) <= 1e-7 * max(1.0, float(np.max(np.abs(b))))
    ref = O.solve(prob["P"], prob["q"], prob["A"], prob["b"], util.oracle_cones(prob["sets"]),
                  O.Settings(max_iter=iters, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9, kkt_solver="cg", tol_constant=1e-10, tol_exponent=0.0))
    for a, b in ((r1.x, ref.x), (r1.s, ref.s), (r1.y, ref.y)):
        assert np.max(np.abs(a - b)) <= 1e-7 * max(1.0, float(np.max(np.abs(b))))
    # captured chain (odd requested length: rounded up) vs direct launches, default schedule with rho adaptation
    monkeypatch.setenv("COSMO_HIP_FOLD_FACTOR", "1")
    monkeypatch.setenv("COSMO_HIP_OP_FOLD", "1")
    res = {}
    for graph in ("0", "1"):
        monkeypatch.setenv("COSMO_HIP_CG_GRAPH", graph)
        monkeypatch.setenv("COSMO_HIP_CG_GRAPH_LEN", "3")
        st = cj.Settings(max_iter=90, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9)
        md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], st)
        res[graph] = cj.optimize(md)
        assert md.handle.fold_stats()["factored_rows"] > 0
    a, b = res["0"], res["1"]
    assert np.array_equal(a.x, b.x) and np.array_equal(a.s, b.s) and np.array_equal(a.y, b.y) and a.kkt_iters_total == b.kkt_iters_total
    assert list(a.info.rho_updates) == list(b.info.rho_updates)
