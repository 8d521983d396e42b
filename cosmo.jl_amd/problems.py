"""Seeded synthetic problem generators for the BASELINE.json configurations.

Every generator returns the problem in COSMO's INTERNAL form  min 1/2 x'Px + q'x  s.t.  A x + s = b, s in K
(what `set!` takes, src/interface.jl:218-250) as a dict {P, q, A, b, sets, name}.  Inputs are generated with
NumPy's `default_rng(seed)` (Julia's MersenneTwister streams cannot be reproduced), so oracle, CPU baseline and
GPU all read identical bytes.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.sparse as sp

from .model import Box, Nonnegatives, PsdConeTriangle, SecondOrderCone, ZeroSet


def _coo_random(rng, m, n, nnz, scale=1.0):
    i = rng.integers(0, m, size=nnz)
    j = rng.integers(0, n, size=nnz)
    v = rng.standard_normal(nnz) * scale
    M = sp.coo_matrix((v, (i, j)), shape=(m, n)).tocsc()   # duplicates summed
    M.sort_indices()
    return M


def dense_qp(n=200, half_m=150, seed=1):
    """cfg1: the structure of examples/qp.jl:13-22 at n=200, m=300: P = Q' diag(U[0.1,2]) Q
    (test/UnitTests/COSMOTestUtils.jl:11-19), one Nonnegatives(2*half_m) constraint [-G; G] x + [u; -l] >= 0."""
    rng = np.random.default_rng(seed)
    Q = np.linalg.qr(rng.standard_normal((n, n)))[0]
    P = Q.T @ np.diag(rng.uniform(0.1, 2.0, n)) @ Q
    P = (P + P.T) / 2
    q = rng.standard_normal(n)
    G = rng.standard_normal((half_m, n))
    x0 = rng.standard_normal(n)
    l = G @ x0 - rng.uniform(size=half_m)
    u = G @ x0 + rng.uniform(size=half_m)
    Ac = np.vstack([-G, G]); bc = np.concatenate([u, -l])
    return dict(name="cfg1_dense_qp", P=sp.csc_matrix(P), q=q, A=sp.csc_matrix(-Ac), b=bc, sets=[Nonnegatives(2 * half_m)])


def sparse_box_qp(n=100_000, m=200_000, nnz=2_000_000, seed=2, eq_frac=0.10, loose_frac=0.05):
    """cfg2: random sparse QP, Box cone: user-level Constraint(G, 0, Box(l,u)) => internal A = -G, b = 0, s = G x."""
    rng = np.random.default_rng(seed)
    G = _coo_random(rng, m, n, nnz)
    S = _coo_random(rng, n, n, 2 * n, 0.1)
    S = (S + S.T).tocsc()
    rowsum = np.asarray(abs(S).sum(axis=1)).ravel()
    P = (S + sp.diags(rowsum + rng.uniform(0.1, 1.0, n))).tocsc()
    P.sort_indices()
    q = rng.standard_normal(n)
    x0 = rng.standard_normal(n)
    c = G @ x0
    l = c - np.abs(rng.standard_normal(m))
    u = c + np.abs(rng.standard_normal(m))
    kind = rng.uniform(size=m)
    eq = kind < eq_frac
    loose = (kind >= eq_frac) & (kind < eq_frac + loose_frac)
    l[eq] = c[eq]; u[eq] = c[eq]
    l[loose] = -1e30; u[loose] = 1e30
    A = (-G).tocsc(); A.sort_indices()
    return dict(name="cfg2_sparse_box_qp", P=P, q=q, A=A, b=np.zeros(m), sets=[Box(l, u)])


def socp(n=500, m=1000, ncones=50, nnz=10_000, seed=1000):
    """cfg3 (one problem of the batch): strictly feasible SOCP with `ncones` SecondOrderCone(m/ncones)."""
    rng = np.random.default_rng(seed)
    k = m // ncones
    A = _coo_random(rng, m, n, nnz)
    p = 0.01 + rng.uniform(size=n)
    P = sp.diags(p).tocsc()
    x0 = rng.standard_normal(n)
    s0 = np.empty(m); y0 = np.empty(m)
    for c in range(ncones):
        v = rng.standard_normal(k - 1)
        s0[c * k] = np.linalg.norm(v) + 1.0; s0[c * k + 1:(c + 1) * k] = v
        v = rng.standard_normal(k - 1)
        y0[c * k] = np.linalg.norm(v) + 1.0; y0[c * k + 1:(c + 1) * k] = v
    b = A @ x0 + s0
    q = -(P @ x0) - A.T @ y0
    return dict(name="cfg3_socp", P=P, q=q, A=A, b=b, sets=[SecondOrderCone(k) for _ in range(ncones)])


def svec(M):
    """Symmetric matrix -> scaled upper triangle, column by column (src/convexset.jl:462-472)."""
    d = M.shape[0]
    jj, ii = np.tril_indices(d)
    out = M[ii, jj] * math.sqrt(2.0)
    out[ii == jj] = M[ii[ii == jj], jj[ii == jj]]
    return out


def smat(x):
    d = (math.isqrt(1 + 8 * x.size) - 1) // 2
    M = np.zeros((d, d))
    jj, ii = np.tril_indices(d)
    v = x / math.sqrt(2.0)
    v[ii == jj] = x[ii == jj]
    M[ii, jj] = v; M[jj, ii] = v
    return M


def closest_correlation(d=2000, seed=4):
    """cfg4: min 1/2 ||X - C||_F^2, X_ii = 1, X psd with x = svec(X) (structure of test/UnitTests/closestcorr.jl:41-63,
    triangle cone).  Internal rows: ZeroSet(d) (diagonal picks) then PsdConeTriangle(d(d+1)/2) (A = -I)."""
    rng = np.random.default_rng(seed)
    G = rng.uniform(-1.0, 1.0, size=(d, d))
    Cm = (G + G.T) / 2
    nt = d * (d + 1) // 2
    diag_idx = np.array([(j + 1) * (j + 2) // 2 - 1 for j in range(d)])
    A1 = sp.csc_matrix((np.ones(d), (np.arange(d), diag_idx)), shape=(d, nt))
    # Constraint(A1, -1, ZeroSet) -> internal -A1, b = -1 ; Constraint(I, 0, PsdConeTriangle) -> internal -I, b = 0
    A = sp.vstack([-A1, -sp.identity(nt, format="csc")], format="csc")
    A.sort_indices()
    b = np.concatenate([-np.ones(d), np.zeros(nt)])
    return dict(name="cfg4_closest_correlation", P=sp.identity(nt, format="csc"), q=-svec(Cm), A=A, b=b,
                sets=[ZeroSet(d), PsdConeTriangle(nt)], C=Cm)


def chordal_sdp(ncliques=400, dmin=20, dmax=200, sep_min=2, sep_max=12, n_total=50_000, n_zero=1000, n_nonneg=5000,
                seed=5):
    """cfg5: an ALREADY-DECOMPOSED sparse SDP, i.e. what `augment_clique_based!` emits
    (src/chordal_decomposition/transformations.jl:152-200): a random clique tree; one PsdConeTriangle per clique; one
    extra variable column with a (+1 child row, -1 parent row) pair per overlapped svec entry (:319-344); plus
    ZeroSet/Nonnegatives rows with 10 nnz per row.  Feasible by construction (b = A x0 + s0, s0 in K)."""
    rng = np.random.default_rng(seed)
    dk = rng.integers(dmin, dmax + 1, size=ncliques)
    parent = np.full(ncliques, -1)
    sep = np.zeros(ncliques, dtype=np.int64)
    for k in range(1, ncliques):
        parent[k] = rng.integers(0, k)
        sep[k] = min(rng.integers(sep_min, sep_max + 1), dk[k] - 1, dk[parent[k]] - 1)
    n_overlap = int(sum(s * (s + 1) // 2 for s in sep))
    n0 = n_total - n_overlap
    if n0 <= 0:
        raise ValueError("n_total too small for the overlaps")
    tri = dk * (dk + 1) // 2
    off = np.concatenate([[0], np.cumsum(tri)])
    m_psd = int(off[-1])
    rows, cols, vals = [], [], []
    # original variables: every svec entry of every clique sees ~1 original variable (sparse selection)
    r = np.arange(m_psd)
    c = rng.integers(0, n0, size=m_psd)
    rows.append(r); cols.append(c); vals.append(rng.standard_normal(m_psd))
    # consensus columns
    col = n0
    def svec_index(i, j):  # i <= j, 0-based, column-major upper
        return j * (j + 1) // 2 + i
    for k in range(1, ncliques):
        p = parent[k]
        ck = rng.choice(dk[k], size=sep[k], replace=False); ck.sort()
        cp = rng.choice(dk[p], size=sep[k], replace=False); cp.sort()
        for a in range(sep[k]):
            for bb in range(a, sep[k]):
                rows.append(np.array([off[k] + svec_index(ck[a], ck[bb]), off[p] + svec_index(cp[a], cp[bb])]))
                cols.append(np.array([col, col])); vals.append(np.array([1.0, -1.0]))
                col += 1
    assert col == n_total
    Apsd = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(m_psd, n_total)).tocsc()
    Az = _coo_random(rng, n_zero, n_total, 10 * n_zero)
    An = _coo_random(rng, n_nonneg, n_total, 10 * n_nonneg)
    A = sp.vstack([Az, An, Apsd], format="csc"); A.sort_indices()
    x0 = rng.standard_normal(n_total)
    s0 = np.concatenate([np.zeros(n_zero), rng.uniform(0.1, 1.0, n_nonneg)] +
                        [svec((lambda B: B @ B.T / d + 0.1 * np.eye(d))(rng.standard_normal((d, d)))) for d in dk])
    b = A @ x0 + s0
    # dual-feasible interior point y0 (free on ZeroSet, > 0 on Nonnegatives, PD on every clique) => bounded problem
    y0 = np.concatenate([rng.standard_normal(n_zero), rng.uniform(0.1, 1.0, n_nonneg)] +
                        [svec((lambda B: B @ B.T / d + 0.1 * np.eye(d))(rng.standard_normal((d, d)))) for d in dk])
    P = sp.diags(rng.uniform(0.5, 1.5, n_total)).tocsc()
    q = -(P @ x0) - A.T @ y0
    sets = [ZeroSet(n_zero), Nonnegatives(n_nonneg)] + [PsdConeTriangle(int(t)) for t in tri]
    return dict(name="cfg5_chordal_sdp", P=P, q=q, A=A, b=b, sets=sets, clique_dims=dk)
