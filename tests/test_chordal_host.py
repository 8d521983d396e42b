"""CPU tests of libcosmo_chordal.so (chordal decomposition front-end restated in C++).  Goldens of the
reference's own DecompositionTests for the merging machinery, structural properties of the decomposition, and the
decomposed == undecomposed equivalence of test/UnitTests/DecompositionTests/chordal_decomposition_triangle.jl solved with the
CPU oracle.  (The elimination ordering comes from the external AMD in the reference -- parity unpinned for that step; every
test below is ordering independent or passes the ordering explicitly.)"""
import itertools
import math

import numpy as np
import pytest
import scipy.sparse as sp

from cosmo_jl_amd import _chordal as ch
from oracle import cosmo_oracle as O

PSD_TRI, ZERO, NONNEG = 5, 0, 1


# ---- test/UnitTests/DecompositionTests/clique_merging_example.jl ------------------------------------------------------------
SND = [{15, 16, 17}, {5, 9}, {3, 4}, {1}, {2}, {6}, {7, 8}, {12, 13, 14}, {10, 11}]
SEP = [set(), {15, 16}, {5, 15}, {3}, {3, 4}, {9, 16}, {9, 15}, {16, 17}, {13, 14, 17}]
PAR = [0, 1, 2, 3, 3, 2, 2, 1, 8]
POST = list(range(9, 0, -1))


def test_parent_child_merge_golden():
    pairs, dec, num, par = ch.test_merge_tree(SND, SEP, PAR, POST, 17, ch.PARENT_CHILD_MERGE)
    assert pairs.tolist() == [[1, 2], [1, 3], [1, 4], [1, 5], [1, 6], [1, 7], [1, 8], [8, 9]]          # :106
    assert dec.tolist() == [True, True, True, True, True, False, False, True]                          # :108
    assert num == 6                                                                                    # :110
    assert par.tolist() == [0, -1, -1, -1, -1, -1, 1, 1, -1]                                           # cliques 2..6 and 9 merged away, 7 and 8 hang off the root


def test_clique_graph_merge_golden():
    full = [a | b for a, b in zip(SND, SEP)]
    rows, cols, w, _ = ch.test_reduced_clique_graph(full, SEP)
    ref_rows = [2, 8, 3, 6, 7, 4, 5, 5, 9]; ref_cols = [1, 1, 2, 2, 2, 3, 3, 4, 8]                     # get_adjacency_matrix (:52-66)
    ref = {}
    for r, c in zip(ref_rows, ref_cols):
        d1, d2, du = len(full[r - 1]), len(full[c - 1]), len(full[r - 1] | full[c - 1])
        ref[(r, c)] = d1 ** 3 + d2 ** 3 - du ** 3                                                      # complexity_savings (:44-49)
    assert {(int(r), int(c)): float(x) for r, c, x in zip(rows, cols, w)} == ref                       # :127
    pairs, dec, num, par = ch.test_merge_tree(SND, SEP, PAR, POST, 17, ch.CLIQUE_GRAPH_MERGE)
    assert num == 0                                                                                    # :133: every weight is negative
    assert par.tolist() == PAR                                                                         # :134: Kruskal recovers the clique tree


def test_reduced_clique_graph_habib_stacho():
    # test/UnitTests/DecompositionTests/reduced_clique_graph.jl:6-46 (Habib & Stacho 2011, Fig. 1)
    snd = [{4, 5}, {1, 4, 6}, {1, 7}, {1, 8}, {1, 3, 4}, {1, 2, 3}, {2, 3, 9}, {3, 4, 11}, {3, 10}]
    sep = [{1, 3}, {1, 4}, {2, 3}, {3, 4}, {1}, {3}, {4}]
    rows, cols, _, perm = ch.test_reduced_clique_graph(snd, sep)
    edges = list(zip(rows.tolist(), cols.tolist()))
    edges_ref = [(2, 1), (5, 1), (8, 1), (9, 8), (9, 5), (9, 7), (7, 6), (6, 4), (5, 4), (4, 2), (4, 3), (3, 2), (5, 3), (6, 3), (9, 6), (8, 5), (5, 2), (6, 5)]
    assert set(edges) == set(edges_ref) and len(edges) == len(edges_ref)
    permissible_ref = {edges_ref[i - 1] for i in (7, 11, 16, 17, 18)}
    assert {e for e, p in zip(edges, perm) if p} <= permissible_ref and any(perm)


# ---- structure of the decomposition ---------------------------------------------------------------------------------------
def _svec(M):
    d = M.shape[0]
    out = []
    for j in range(d):
        for i in range(j + 1):
            out.append(M[i, j] if i == j else math.sqrt(2) * M[i, j])
    return np.array(out)


def _smat(x):
    d = (math.isqrt(1 + 8 * x.size) - 1) // 2
    M = np.zeros((d, d)); k = 0
    for j in range(d):
        for i in range(j + 1):
            v = x[k] if i == j else x[k] / math.sqrt(2)
            M[i, j] = M[j, i] = v; k += 1
    return M


def _is_clique_cover(pattern, cliques):
    d = pattern.shape[0]
    cov = np.zeros((d, d), dtype=bool)
    for c in cliques:
        idx = np.array(c) - 1
        cov[np.ix_(idx, idx)] = True
    return np.all(cov[pattern != 0])


def test_banded_pattern_gives_window_cliques():
    d, w = 8, 2
    pat = np.array([[1.0 if abs(i - j) <= w else 0.0 for j in range(d)] for i in range(d)])
    A = sp.csc_matrix(_svec(pat)[:, None])
    dec = ch.Decomposition(A, np.zeros(A.shape[0]), [PSD_TRI], [A.shape[0]], merge_strategy=ch.NO_MERGE, orderings=[np.arange(1, d + 1)])
    cl = dec.cliques(1)
    assert sorted(cl) == [list(range(k, k + w + 1)) for k in range(1, d - w + 1)]     # the maximal cliques of a band graph
    assert dec.num_decomposed == 1 and dec.kinds.tolist() == [PSD_TRI] * len(cl)
    assert dec.dims.tolist() == [(w + 1) * (w + 2) // 2] * len(cl)
    assert dec.num_overlaps == (len(cl) - 1) * w * (w + 1) // 2                       # every tree edge shares a w x w block
    assert dec.n_new == 1 + dec.num_overlaps and dec.m_new == int(dec.dims.sum())
    # with the default strategy the merged cliques still cover the pattern and only positive-saving merges happened
    dec2 = ch.Decomposition(A, np.zeros(A.shape[0]), [PSD_TRI], [A.shape[0]])
    assert _is_clique_cover(pat, dec2.cliques(1))
    # minimum-degree ordering instead of the natural one: same band cliques (a band graph is already chordal)
    dec3 = ch.Decomposition(A, np.zeros(A.shape[0]), [PSD_TRI], [A.shape[0]], merge_strategy=ch.NO_MERGE)
    assert sorted(dec3.cliques(1)) == sorted(cl)


def test_dense_and_non_psd_cones_pass_through():
    rng = np.random.default_rng(0)
    A = sp.csc_matrix(rng.normal(size=(6 + 3 + 2, 2)))
    dec = ch.Decomposition(A, rng.normal(size=11), [PSD_TRI, NONNEG, ZERO], [6, 3, 2])
    assert dec.num_decomposed == 0 and dec.kinds.tolist() == [PSD_TRI, NONNEG, ZERO] and dec.dims.tolist() == [6, 3, 2]
    assert (dec.A != A).nnz == 0 and dec.cone_map.tolist() == [1, 2, 3]


@pytest.mark.parametrize("strategy", [ch.NO_MERGE, ch.PARENT_CHILD_MERGE, ch.CLIQUE_GRAPH_MERGE])
def test_random_chordal_patterns_are_valid_decompositions(strategy):
    rng = np.random.default_rng(5 + strategy)
    for trial in range(6):
        d = int(rng.integers(6, 30))
        # random sparse symmetric pattern (not chordal in general: the symbolic factorisation adds the fill)
        M = np.triu((rng.uniform(size=(d, d)) < 0.15).astype(float), 1)
        pat = M + M.T + np.eye(d)
        ncol = 3
        cols = [(_svec((lambda S: (S + S.T) / 2)(pat * rng.normal(size=(d, d))))) for _ in range(ncol)]
        A = sp.csc_matrix(np.array(cols).T)
        b = _svec(pat * 0.5)
        dec = ch.Decomposition(A, b, [PSD_TRI], [A.shape[0]], merge_strategy=strategy)
        if dec.num_decomposed == 0:
            continue
        cl = dec.cliques(1)
        assert _is_clique_cover(pat, cl)
        # running intersection property along the post order: clique_i intersect (union of later cliques) lies in one later clique
        for i in range(len(cl) - 1):
            later = set().union(*[set(c) for c in cl[i + 1:]])
            inter = set(cl[i]) & later
            assert any(inter <= set(c) for c in cl[i + 1:])
        # the transformation keeps every nonzero of A and b exactly once, and adds one (+1, -1) column per overlap entry
        An = dec.A.toarray()
        assert np.isclose(np.abs(An[:, :ncol]).sum(), np.abs(A.toarray()).sum()) and np.isclose(np.abs(dec.b).sum(), np.abs(b).sum())
        ov = An[:, ncol:]
        assert ov.shape[1] == dec.num_overlaps and np.all((ov != 0).sum(axis=0) == 2) and np.all(ov.sum(axis=0) == 0)
        # reverse of a consistent decomposed vector: entries of one PSD matrix S copied into the clique blocks come back as S on the pattern
        S = pat * rng.normal(size=(d, d)); S = (S + S.T) / 2
        s_dec = np.zeros(dec.m_new); off = 0
        for c, dim in zip(cl[::-1], dec.dims):                       # cones are emitted in descending post order
            idx = np.array(c) - 1
            s_dec[off:off + dim] = _svec(S[np.ix_(idx, idx)]); off += dim     # mu semantics: overlapping entries are overwritten
        _, mu = dec.reverse(np.zeros(dec.m_new), s_dec)
        Mrec = _smat(mu)
        cov = np.zeros((d, d), dtype=bool)
        for c in cl:
            idx = np.array(c) - 1; cov[np.ix_(idx, idx)] = True
        assert np.allclose(Mrec[cov], S[cov]) and not Mrec[~cov].any()


# ---- decomposed == undecomposed (chordal_decomposition_triangle.jl:1-150), solved with the CPU oracle ------------------------
def _posdef(rng, d, lo, hi):
    Q = np.linalg.qr(rng.normal(size=(d, d)))[0]
    return (Q * rng.uniform(lo, hi, d)) @ Q.T


def _equivalence_problem(seed):
    rng = np.random.default_rng(seed)
    A1 = rng.uniform(size=(4, 4)); A1 = 0.5 * (A1 + A1.T); A1[0, 2] = A1[0, 3] = A1[2, 0] = A1[3, 0] = 0
    S1 = A1 + (np.linalg.eigvalsh(A1)[0] + 1) * np.eye(4)          # NB as in the reference: feasible slack
    a2 = rng.uniform(size=2)
    A3 = rng.uniform(size=(4, 4)); A3 = 0.5 * (A3 + A3.T)
    for (i, j) in [(1, 3), (0, 2), (1, 2)]:
        A3[i, j] = A3[j, i] = 0
    S3 = A3 + (np.linalg.eigvalsh(A3)[0] + 1) * np.eye(4)
    A4 = rng.uniform(size=(3, 3)); A4 = 0.5 * (A4 + A4.T)
    S4 = A4 + (np.linalg.eigvalsh(A4)[0] + 1) * np.eye(3)
    x = rng.uniform(size=1)
    A = np.concatenate([_svec(A1), a2, _svec(A3), _svec(A4)])[:, None]
    s = np.concatenate([_svec(S1), [0, 0], _svec(S3), _svec(S4)])
    b = A @ x + s
    y = np.concatenate([_svec(_posdef(rng, 4, 0.1, 1.0)), rng.uniform(size=2), _svec(_posdef(rng, 4, 0.1, 1.0)), _svec(_posdef(rng, 3, 0.1, 1.0))])
    q = -(A.T @ y)
    kinds = [PSD_TRI, ZERO, PSD_TRI, PSD_TRI]; dims = [10, 2, 10, 6]
    return sp.csc_matrix(A), b, q, kinds, dims


@pytest.mark.parametrize("strategy", [ch.NO_MERGE, ch.CLIQUE_GRAPH_MERGE])
def test_decomposed_problem_is_equivalent(strategy):
    A, b, q, kinds, dims = _equivalence_problem(144545)
    P = sp.csc_matrix((1, 1))
    cones = [O.Cone(k, d) for k, d in zip(kinds, dims)]
    st = O.Settings(eps_abs=1e-7, eps_rel=1e-7, max_iter=20000)
    r0 = O.solve(P, q, A, b, cones, st)
    dec = ch.Decomposition(A, b, kinds, dims, merge_strategy=strategy)
    if strategy == ch.NO_MERGE:
        assert dec.num_decomposed == 2 and dec.cone_map.tolist() == [1, 1, 2, 3, 3, 3, 4]   # 4x4 chain patterns: cliques {2,3,4},{1,2} and {1,2}? see below
    n_new = dec.n_new
    Pn = sp.block_diag([P, sp.csc_matrix((n_new - 1, n_new - 1))], format="csc")
    qn = np.concatenate([q, np.zeros(n_new - 1)])
    cones_n = [O.Cone(int(k), int(d)) for k, d in zip(dec.kinds, dec.dims)]
    r1 = O.solve(Pn, qn, dec.A, dec.b, cones_n, st)
    assert r0.status == r1.status == "Solved"
    assert abs(r0.obj_val - r1.obj_val) < 1e-4                                           # chordal_decomposition_triangle.jl:142-144
    s_rec, mu_rec = dec.reverse(r1.s, -r1.y, complete_dual=True)
    assert np.max(np.abs(_smat(s_rec[:10]) - _smat(r0.s[:10]))) < 1e-4                   # :145-150
    assert np.max(np.abs(_smat(s_rec[12:22]) - _smat(r0.s[12:22]))) < 1e-4
    assert np.allclose(s_rec[22:], r0.s[22:], atol=1e-4) and np.allclose(s_rec[10:12], 0, atol=1e-6)
    # completed dual variable: Y = -mu is positive semidefinite and agrees with mu on the sparsity pattern
    for lo, hi in ((0, 10), (12, 22)):
        Y = _smat(-mu_rec[lo:hi])
        assert np.linalg.eigvalsh(Y).min() > -1e-5
    # without completion the entries outside the pattern are zero
    _, mu_nc = dec.reverse(r1.s, -r1.y, complete_dual=False)
    assert _smat(mu_nc[0:10])[0, 2] == 0.0 and _smat(mu_nc[0:10])[0, 3] == 0.0


# ---- traditional transformation (compact_transformation = false): s = H sbar, incl. square PsdCone -----------------------
def _equivalence_problem_square(seed):
    """CONFIG 1/2 of chordal_decomposition_triangle.jl:62-83: the same data with PsdCone (vec) instead of PsdConeTriangle (svec)."""
    rng = np.random.default_rng(seed)
    A1 = rng.uniform(size=(4, 4)); A1 = 0.5 * (A1 + A1.T); A1[0, 2] = A1[0, 3] = A1[2, 0] = A1[3, 0] = 0
    S1 = A1 + (np.linalg.eigvalsh(A1)[0] + 1) * np.eye(4)
    a2 = rng.uniform(size=2)
    A3 = rng.uniform(size=(4, 4)); A3 = 0.5 * (A3 + A3.T)
    for (i, j) in [(1, 3), (0, 2), (1, 2)]:
        A3[i, j] = A3[j, i] = 0
    S3 = A3 + (np.linalg.eigvalsh(A3)[0] + 1) * np.eye(4)
    A4 = rng.uniform(size=(3, 3)); A4 = 0.5 * (A4 + A4.T)
    S4 = A4 + (np.linalg.eigvalsh(A4)[0] + 1) * np.eye(3)
    x = rng.uniform(size=1)
    vec = lambda M: M.reshape(-1, order="F")
    A = np.concatenate([vec(A1), a2, vec(A3), vec(A4)])[:, None]
    s = np.concatenate([vec(S1), [0, 0], vec(S3), vec(S4)])
    b = A @ x + s
    y = np.concatenate([vec(_posdef(rng, 4, 0.1, 1.0)), rng.uniform(size=2), vec(_posdef(rng, 4, 0.1, 1.0)), vec(_posdef(rng, 3, 0.1, 1.0))])
    q = -(A.T @ y)
    return sp.csc_matrix(A), b, q, [4, ZERO, 4, 4], [16, 2, 16, 9]


@pytest.mark.parametrize("shape", ["triangle", "square"])
def test_traditional_transformation_is_equivalent(shape):
    if shape == "triangle":
        A, b, q, kinds, dims = _equivalence_problem(144545)
    else:
        A, b, q, kinds, dims = _equivalence_problem_square(144545)
    P = sp.csc_matrix((1, 1))
    cones = [O.Cone(k, d) for k, d in zip(kinds, dims)]
    st = O.Settings(eps_abs=1e-7, eps_rel=1e-7, max_iter=20000)
    r0 = O.solve(P, q, A, b, cones, st)
    dec = ch.Decomposition(A, b, kinds, dims, merge_strategy=ch.NO_MERGE, compact=False)
    # square cones: the reference flags the positions i^2 instead of the diagonal (kept, see chordal_api.cpp), which fills the
    # (1,3)/(4,1) zeros of the first 4x4 block, so only the second block decomposes there
    assert dec.num_decomposed == (2 if shape == "triangle" else 1)
    m0 = A.shape[0]
    assert dec.kinds[0] == ZERO and dec.dims[0] == m0 and dec.cone_map[0] == 0          # the ZeroSet(m) block of the augmented system
    assert dec.m_new == m0 + dec.num_overlaps and dec.n_new == 1 + dec.num_overlaps
    An = dec.A.toarray()
    assert np.array_equal(An[:m0, :1], A.toarray()) and np.array_equal(An[m0:, 1:], -np.eye(dec.num_overlaps))      # [A H; 0 -I]
    H = An[:m0, 1:]
    assert set(np.unique(H)) <= {0.0, 1.0} and np.all(H.sum(axis=0) == 1)
    Pn = sp.block_diag([P, sp.csc_matrix((dec.n_new - 1, dec.n_new - 1))], format="csc")
    r1 = O.solve(Pn, np.concatenate([q, np.zeros(dec.n_new - 1)]), dec.A, dec.b, [O.Cone(int(k), int(d)) for k, d in zip(dec.kinds, dec.dims)], st)
    assert r0.status == r1.status == "Solved" and abs(r0.obj_val - r1.obj_val) < 1e-4
    s_rec, mu_rec = dec.reverse(r1.s, -r1.y, complete_dual=True)
    assert np.max(np.abs(s_rec - r0.s)) < 1e-4
    lo = 0 if shape == "triangle" else 18
    d0 = dims[0]
    Y = _smat(-mu_rec[:d0]) if shape == "triangle" else (-mu_rec[lo:lo + 16]).reshape(4, 4, order="F")
    Y = (np.triu(Y) + np.triu(Y, 1).T)
    assert np.linalg.eigvalsh(Y).min() > -1e-5
    # compact transformation leaves square cones alone (the reference's add_entries! is specialised on PsdConeTriangle)
    if shape == "square":
        assert ch.Decomposition(A, b, kinds, dims, merge_strategy=ch.NO_MERGE, compact=True).num_decomposed == 0


# ---- the reference's literal LMI instance (test/UnitTests/nuclear_norm_minimization.jl:17-46) -----------------------------------
def _sigma_max_problem():
    """min t  s.t.  Y21 <= 4, Y22 >= 3, sum(Y) >= 12, [[t I, Y], [Y', t I]] psd  with x = [t; vec(Y)], Y 3 x 3 -- the Convex.jl
    problem "sdp_sigma_max_atom" that once broke the compact transformation.  Returns internal (A, b) and cones (user form A x + b in K)."""
    q = np.concatenate([[1.0], np.zeros(9)])
    r1 = np.zeros((1, 10)); r1[0, 2] = -1.0
    r2 = np.zeros((1, 10)); r2[0, 5] = 1.0
    r3 = np.concatenate([[0.0], np.ones(9)])[None, :]
    a1 = np.array([-1.0, 0, -1, 0, 0, -1, 0, 0, 0, -1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 0, -1])
    a2 = np.zeros((21, 9))
    for row, col in ((7, 1), (8, 2), (9, 3), (11, 4), (12, 5), (13, 6), (16, 7), (17, 8), (18, 9)):   # 1-based in the reference
        a2[row - 1, col - 1] = -np.sqrt(2.0)
    A_lmi = np.hstack([a1[:, None], a2])
    cons = [(r1, np.array([4.0]), O.NONNEG, 1), (r2, np.array([-3.0]), O.NONNEG, 1), (r3, np.array([-12.0]), O.NONNEG, 1), (-A_lmi, np.zeros(21), O.PSD_TRIANGLE, 21)]
    return q, cons


@pytest.mark.parametrize("compact", [True, False])
def test_sigma_max_lmi_golden_through_the_decomposition(compact):
    q, cons = _sigma_max_problem()
    A, b, cones = O.assemble([O.Constraint(Ai, bi, O.Cone(k, d, constr_type=(np.zeros(d, dtype=bool) if k == O.NONNEG else None))) for (Ai, bi, k, d) in cons])
    kinds = [c.kind for c in cones]; dims = [c.dim for c in cones]
    dec = ch.Decomposition(A, b, kinds, dims, compact=compact)
    assert dec.num_decomposed == 1                                   # the 6 x 6 block-arrow LMI is decomposed
    n_new = dec.n_new
    P = sp.csc_matrix((n_new, n_new)); qn = np.concatenate([q, np.zeros(n_new - 10)])
    cones_n = [O.Cone(int(k), int(d), constr_type=(np.zeros(int(d), dtype=bool) if int(k) == O.NONNEG else None)) for k, d in zip(dec.kinds, dec.dims)]
    r = O.solve(P, qn, dec.A, dec.b, cones_n, O.Settings(eps_abs=1e-5, eps_rel=1e-5))
    assert r.status == "Solved"
    Y = r.x[1:10].reshape(3, 3, order="F"); t = r.x[0]
    assert Y[1, 0] <= 4 + 1e-3 and Y[1, 1] >= 3 - 1e-3 and Y.sum() - 12.0 >= -1e-3     # nuclear_norm_minimization.jl:39-41
    assert abs(np.linalg.svd(Y, compute_uv=False).max() - t) <= 1e-3                   # :43-45
    # and the undecomposed problem gives the same optimum
    r0 = O.solve(sp.csc_matrix((10, 10)), q, A, b, cones, O.Settings(eps_abs=1e-5, eps_rel=1e-5))
    assert r0.status == "Solved" and abs(r0.x[0] - t) <= 2e-3
