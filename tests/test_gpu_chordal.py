"""GPU tests of the chordal decomposition front-end in the loop: a sparse SDP solved on the device with and
without decomposition (test/UnitTests/DecompositionTests/chordal_decomposition_triangle.jl:142-150 asserts exactly this
equivalence), with every merge strategy, and the reassembled / completed variables."""
import math

import numpy as np
import pytest
import scipy.sparse as sp

import cosmo_jl_amd as cj
from tests.test_chordal_host import _equivalence_problem, _smat, _svec

pytestmark = pytest.mark.gpu
F = cj._ffi


def _sets(kinds, dims):
    m = {F.PSD_TRIANGLE: cj.PsdConeTriangle, F.ZERO: cj.ZeroSet, F.NONNEG: cj.Nonnegatives}
    return [m[k](d) for k, d in zip(kinds, dims)]


@pytest.mark.parametrize("strategy", [cj.NoMerge, cj.ParentChildMerge, cj.CliqueGraphMerge])
def test_reference_equivalence_problem(strategy):
    A, b, q, kinds, dims = _equivalence_problem(144545)
    P = sp.csc_matrix((1, 1))
    res = {}
    for dec in (False, True):
        model = cj.Model()
        model.set(P, q, A, b, _sets(kinds, dims), cj.Settings(decompose=dec, merge_strategy=strategy, complete_dual=True, eps_abs=1e-6, eps_rel=1e-6,
                                                                kkt_solver=cj.with_options(cj.CGIndirectKKTSolver, tol_constant=1e-9, tol_exponent=0.0)))
        res[dec] = (cj.optimize(model), model)
    r0, _ = res[False]; r1, m1 = res[True]
    assert r0.status == r1.status == "Solved"
    assert (m1.chordal is not None) == (strategy is not cj.ParentChildMerge or m1.chordal is not None)
    assert abs(r0.obj_val - r1.obj_val) < 1e-4
    assert r1.x.size == 1 and r1.s.size == r0.s.size
    assert np.max(np.abs(_smat(r1.s[:10]) - _smat(r0.s[:10]))) < 1e-3 and np.max(np.abs(_smat(r1.s[12:22]) - _smat(r0.s[12:22]))) < 1e-3
    for lo, hi in ((0, 10), (12, 22)):
        assert np.linalg.eigvalsh(_smat(r1.y[lo:hi])).min() > -1e-4          # completed dual variable is PSD


def test_banded_sdp_decomposed_matches_undecomposed():
    # min <C, X> s.t. X_ii = 1, X >= 0 with a banded C (bandwidth 4, d = 60): the sparsity pattern of the slack is the band
    rng = np.random.default_rng(3)
    d, w = 60, 4
    band = np.array([[1.0 if abs(i - j) <= w else 0.0 for j in range(d)] for i in range(d)])
    Cm = band * rng.normal(size=(d, d)); Cm = (Cm + Cm.T) / 2
    # dual form: variables y (d), constraint  C - diag(y) in PSD  <=>  A y + s = b with A = svec(e_i e_i'), b = svec(C); maximise sum(y)
    nt = d * (d + 1) // 2
    diag_idx = np.array([(j + 1) * (j + 2) // 2 - 1 for j in range(d)])
    A = sp.csc_matrix((np.ones(d), (diag_idx, np.arange(d))), shape=(nt, d))
    b = _svec(Cm)
    q = -np.ones(d)
    P = sp.csc_matrix((d, d))
    out = {}
    for dec in (False, True):
        model = cj.Model(); model.set(P, q, A, b, [cj.PsdConeTriangle(nt)], cj.Settings(decompose=dec, eps_abs=1e-4, eps_rel=1e-4, max_iter=20000,
                                                                                          kkt_solver=cj.with_options(cj.CGIndirectKKTSolver, tol_constant=1e-9, tol_exponent=0.0)))
        out[dec] = (cj.optimize(model), model)
    r0, _ = out[False]; r1, m1 = out[True]
    assert m1.chordal is not None and m1.chordal.num_decomposed == 1
    cl = m1.chordal.cliques(1)
    assert 1 < len(cl) <= d - w and max(len(c) for c in cl) < d
    assert r0.status == r1.status == "Solved"
    assert abs(r0.obj_val - r1.obj_val) < 5e-3 * (1 + abs(r0.obj_val))
    assert np.linalg.norm(r0.x - r1.x) < 5e-2 * (1 + np.linalg.norm(r0.x))
    S = _smat(r1.s)
    assert np.linalg.eigvalsh(S).min() > -1e-2 and np.allclose(S[band == 0], 0.0, atol=1e-3)   # merged cliques carry fill entries that the solve drives to 0
    print("banded SDP: iterations undecomposed %d, decomposed %d (%d cliques)" % (r0.iter, r1.iter, len(cl)))


def test_traditional_transformation_on_device():
    """compact_transformation = false (s = H sbar with the ZeroSet(m) block): same optimum as the undecomposed solve."""
    A, b, q, kinds, dims = _equivalence_problem(144545)
    P = sp.csc_matrix((1, 1))
    tight = cj.with_options(cj.CGIndirectKKTSolver, tol_constant=1e-9, tol_exponent=0.0)
    out = {}
    for dec in (False, True):
        model = cj.Model()
        model.set(P, q, A, b, _sets(kinds, dims), cj.Settings(decompose=dec, compact_transformation=False, merge_strategy=cj.NoMerge, complete_dual=True,
                                                                eps_abs=1e-6, eps_rel=1e-6, kkt_solver=tight))
        out[dec] = (cj.optimize(model), model)
    r0, _ = out[False]; r1, m1 = out[True]
    assert m1.chordal is not None and not m1.chordal.compact and m1.sets[0].kind == F.ZERO and m1.sets[0].dim == A.shape[0]
    assert r0.status == r1.status == "Solved" and abs(r0.obj_val - r1.obj_val) < 1e-4
    assert r1.s.size == r0.s.size and np.max(np.abs(r1.s - r0.s)) < 1e-3
