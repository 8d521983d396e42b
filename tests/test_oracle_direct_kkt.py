"""The direct-KKT leg of the compiled CPU baseline (oracle/cosmo_oracle_c.c: ldl_etree / ldl_factor / ldl_solve, a restatement of the published
QDLDL algorithm behind the reference's DEFAULT solver QdldlKKTSolver, /root/reference/src/linear_solver/kktsolver.jl:285-320).  QDLDL.jl ("0.4.1")
and AMD.jl are not vendored in the reference tree, so the restatement is PARITY UNPINNED against the package; it is anchored here on
  (a) the dense solve of the same quasi-definite system  [P + sigma I, A'; A, -diag(1/rho)] sol = ls   (what `solve!` must return),
  (b) the NumPy oracle's "qdldl" path (SciPy factorisation) through the WHOLE loop incl. rho updates = refactorisations,
  (c) the inertia test of the constructor (kktsolver.jl:304) and the reference's own error for a non-convex objective,
  (d) permutation invariance and the fill counts of the elimination-tree pass against a dense symbolic factorisation."""
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

import cosmo_jl_amd as cj
from oracle import cosmo_oracle as O
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def OC():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    from oracle import cosmo_oracle_c
    return cosmo_oracle_c


def _ws(prob, kkt="cg", **kw):
    return O.Workspace(prob["P"], prob["q"], prob["A"], prob["b"], util.oracle_cones(prob["sets"]), O.Settings(kkt_solver=kkt, **kw))


def test_one_iteration_equals_the_dense_kkt_solve(OC):
    """max_iter = 0 leaves only the init step (solver.jl:137-138): w = w0 + alpha ([x_tl ; s_tl] - [x0 ; s0]) with [x_tl ; nu] the solution of ONE KKT
    system -- recomputed here with a dense solve of the assembled matrix."""
    rng = np.random.default_rng(3)
    prob = util.random_qp(rng, 40, 5, 30, 20, soc_dims=(4, 6), p_shift=1.0)
    ws = _ws(prob, max_iter=1, scaling=0, check_termination=10 ** 6)
    perm = OC.kkt_ordering(ws, cache=False)
    n, m = ws.n, ws.m
    K = OC.kkt_full(ws).toarray()
    assert np.allclose(K, K.T)
    # the right-hand side of the init step (admm_x!, solver.jl:50-51) from the zero start: w = 0, s = 0
    ls = np.concatenate([-ws.q, ws.b])
    sol = np.linalg.solve(K, ls)
    out = OC.run(ws, direct=dict(perm=perm))
    # after the init step and ONE iteration the loop has solved twice; check the first solve through x of the init step: w_x = alpha * x_tl
    # (w_prev of iteration 1 is what the run returns as x: solver.jl:151, 167-171)
    assert np.max(np.abs(out["x_scaled"] - ws.st.alpha * sol[:n])) <= 1e-11 * max(1.0, np.max(np.abs(sol)))
    assert out["ldl"]["n_solve"] == 2 and out["ldl"]["n_factor"] == 1 and out["ldl"]["positive_D"] == n


@pytest.mark.parametrize("seed", [1, 2])
def test_direct_loop_equals_the_numpy_oracles_direct_path(OC, seed):
    rng = np.random.default_rng(seed)
    prob = util.random_qp(rng, 70, 6, 50, 60, soc_dims=(5, 9, 3), psd_tri_dims=(4, 7), p_shift=3.0)
    ref = _ws(prob, kkt="qdldl", max_iter=3000).optimize()
    ws = _ws(prob, max_iter=3000)
    out = OC.run(ws, direct=dict(perm=OC.kkt_ordering(ws, cache=False)))
    assert out["status"] == ref.status == "Solved" and out["iter"] == ref.iter
    assert len(out["rho_updates"]) == len(ref.rho_updates) >= 2                     # at least one refactorisation happened ...
    assert out["ldl"]["n_factor"] == len(ref.rho_updates)                            # ... one per rho value (update_rho!, kktsolver.jl:316-320)
    np.testing.assert_allclose(out["rho_updates"], ref.rho_updates, rtol=1e-8)
    assert np.max(np.abs(out["x"] - ref.x)) <= 1e-9 * max(1.0, np.max(np.abs(ref.x)))
    assert np.max(np.abs(out["s"] - ref.s)) <= 1e-9 * max(1.0, np.max(np.abs(ref.s)))
    assert abs(out["obj_val"] - ref.obj_val) <= 1e-9 * max(1.0, abs(ref.obj_val))
    assert out["cg_iters_total"] == 0


def test_any_symmetric_permutation_gives_the_same_iterates_and_the_fill_count_is_the_symbolic_one(OC):
    rng = np.random.default_rng(9)
    prob = util.random_qp(rng, 30, 3, 25, 10, p_shift=2.0)
    runs = []
    for k in range(3):
        ws = _ws(prob, max_iter=200, eps_abs=0.0, eps_rel=0.0)
        N = ws.n + ws.m
        perm = OC.kkt_ordering(ws, cache=False) if k == 0 else np.random.default_rng(k).permutation(N).astype(np.int64)
        out = OC.run(ws, direct=dict(perm=perm))
        runs.append(out)
        # fill of L: dense symbolic Cholesky of the permuted pattern
        K = (OC.kkt_full(ws).toarray() != 0)[np.ix_(perm, perm)]
        fill = 0
        for j in range(N):
            below = np.where(K[j + 1:, j])[0] + j + 1
            fill += below.size
            if below.size:
                K[np.ix_(below, below)] = True
        assert out["ldl"]["nnz_L"] == fill == OC.ldl_nnz(ws, perm)
    for o in runs[1:]:
        assert np.max(np.abs(o["x"] - runs[0]["x"])) <= 1e-9 * max(1.0, np.max(np.abs(runs[0]["x"])))
    assert runs[0]["ldl"]["nnz_L"] <= min(o["ldl"]["nnz_L"] for o in runs[1:])       # the minimum-degree ordering fills least


def test_nonconvex_objective_is_the_references_error_and_the_fill_cap_stops_the_run(OC):
    prob = util.random_qp(np.random.default_rng(4), 20, 2, 10, 5)
    P = prob["P"].tolil(); P[0, 0] = -5.0; prob["P"] = P.tocsc()                       # indefinite P: n positive pivots are impossible
    ws = _ws(prob, scaling=0)
    with pytest.raises(ValueError, match="not convex"):                                # kktsolver.jl:304
        OC.run(ws, direct=dict(perm=OC.kkt_ordering(ws, cache=False)))
    prob = util.random_qp(np.random.default_rng(5), 20, 2, 10, 5)
    ws = _ws(prob)
    perm = OC.kkt_ordering(ws, cache=False)
    assert OC.ldl_nnz(ws, perm, cap=3) == -2
    with pytest.raises(OverflowError):
        OC.run(ws, direct=dict(perm=perm, nnz_cap=3))


def test_committed_ordering_of_baseline_config_5_matches_its_pattern(OC):
    """bench.py's direct-KKT leg reads the minimum-degree ordering of BASELINE config 5 from oracle/kkt_perm/ (SuperLU needs ~1 min to hand it out): the
    committed file must belong to the generator's pattern and give the committed fill."""
    prob = cj.problems.chordal_sdp()
    ws = _ws(prob, max_iter=1)
    path = os.path.join(OC.PERM_DIR, "perm_%s.npz" % OC._pattern_key(ws.P, ws.A))
    assert os.path.exists(path), "run `python -c 'import bench; ...'` / OC.kkt_ordering once to regenerate"
    perm = OC.kkt_ordering(ws)
    assert np.array_equal(np.sort(perm), np.arange(ws.n + ws.m))
    assert OC.ldl_nnz(ws, perm) == 15226798
