"""GPU tests of the complex Hermitian PSD cone (PsdConeTriangle{T, Complex{T}}, src/convexset.jl:345-490): the
device projects the real symmetric embedding [[A, -B], [B, A]] with the matrix-sign iteration (csrc/psd_polar.hip)."""
import math

import numpy as np
import pytest
import scipy.sparse as sp

import cosmo_jl_amd as cj
from oracle import cosmo_oracle as O
from tests import util
from tests.util import EPS

pytestmark = pytest.mark.gpu
F = cj._ffi


def _herm(rng, d, lam=None):
    G = rng.normal(size=(d, d)) + 1j * rng.normal(size=(d, d))
    Q = np.linalg.qr(G)[0]
    if lam is None:
        npos = rng.integers(0, d + 1)
        lam = np.concatenate([rng.uniform(0.1, 2, npos), -rng.uniform(0.1, 2, d - npos)])
    H = (Q * lam) @ Q.conj().T
    return (H + H.conj().T) / 2


@pytest.mark.parametrize("dims", [[2, 3, 5, 8], [17, 30, 33], [64, 100, 128], [150]])
def test_complex_projection_matches_oracle(dims):
    rng = np.random.default_rng(sum(dims))
    mats = [_herm(rng, d) for d in dims]
    sets = [cj.ComplexPsdConeTriangle(d * d) for d in dims]
    xs = []
    for H in mats:
        x = np.zeros(H.shape[0] ** 2); O.extract_upper_triangle_complex(H, x); xs.append(x)
    s = np.concatenate(xs)
    m = s.size
    h = cj.Handle(0)
    h.set_problem(sp.identity(2, format="csc"), np.zeros(2), sp.csc_matrix((m, 2)), np.zeros(m))
    h.set_cones([K.kind for K in sets], [K.dim for K in sets], None, None)
    ref = s.copy(); info = {}
    O.project(ref, util.oracle_cones(sets), info)
    out, ranks, _ = h.project(s)
    h.close()
    off = 0
    for K, H, rk, rk_ref in zip(sets, mats, ranks, info["psd_rank"]):
        d = H.shape[0]
        err = np.linalg.norm(out[off:off + K.dim] - ref[off:off + K.dim])
        assert err <= 16 * (2 * d) * EPS * np.linalg.norm(H), (d, err)
        assert rk == rk_ref, (d, rk, rk_ref)
        off += K.dim


def test_complex_special_cases_and_mixed_composite():
    rng = np.random.default_rng(9)
    d = 12
    H = _herm(rng, d)
    B = rng.normal(size=(d, d)) + 1j * rng.normal(size=(d, d))
    mats = [B @ B.conj().T + 0.1 * np.eye(d), -(B @ B.conj().T + 0.1 * np.eye(d)), np.zeros((d, d), dtype=complex), H]
    kinds = [F.PSD_TRIANGLE_COMPLEX] * 4 + [F.NONNEG, F.PSD_TRIANGLE, F.PSD_TRIANGLE_COMPLEX]
    dims = [d * d] * 4 + [5, 10 * 11 // 2, 1]
    xs = []
    for M in mats:
        x = np.zeros(d * d); O.extract_upper_triangle_complex(M, x); xs.append(x)
    xs += [rng.normal(size=5), cj.problems.svec((lambda G: (G + G.T) / 2)(rng.normal(size=(10, 10)))), np.array([-0.3])]
    s = np.concatenate(xs)
    m = s.size
    h = cj.Handle(0)
    h.set_problem(sp.identity(2, format="csc"), np.zeros(2), sp.csc_matrix((m, 2)), np.zeros(m))
    h.set_cones(kinds, dims, None, None)
    cones = [O.Cone(k, dd, constr_type=(np.zeros(dd, dtype=bool) if k == O.NONNEG else None)) for k, dd in zip(kinds, dims)]
    ref = s.copy(); O.project(ref, cones)
    out, ranks, _ = h.project(s)
    h.close()
    assert np.linalg.norm(out - ref) <= 1e-12 * max(1.0, np.linalg.norm(s))
    assert np.array_equal(out[:d * d], s[:d * d]) or np.linalg.norm(out[:d * d] - s[:d * d]) < 1e-12 * np.linalg.norm(s[:d * d])   # PSD input: identity map
    assert not out[2 * d * d:3 * d * d].any() and np.linalg.norm(out[d * d:2 * d * d]) < 1e-12 * np.linalg.norm(s[d * d:2 * d * d])
    assert out[-1] == 0.0                                    # 1 x 1 complex cone: max(x, 0)


def test_least_eigenvalue_golden_on_device():
    # test/UnitTests/least_eigenvalue.jl:32-37
    X = np.array([[1, 1j, 0], [-1j, 1, 1j], [0, -1j, 1]])
    d = 3; n = d * d
    vec_c = np.zeros(n); O.extract_upper_triangle_complex(X, vec_c)
    id_vec = np.zeros(n); id_vec[[k * (k + 1) // 2 - 1 for k in range(1, d + 1)]] = 1.0
    model = cj.Model()
    cj.assemble(model, np.zeros((n, n)), vec_c, [cj.Constraint(id_vec[None, :], [-1.0], cj.ZeroSet), cj.Constraint(np.eye(n), np.zeros(n), cj.ComplexPsdConeTriangle(n))],
                settings=cj.Settings(eps_abs=1e-5, eps_rel=1e-5))
    res = cj.optimize(model)
    assert res.status == "Solved" and abs(res.obj_val - (1 - math.sqrt(2))) < 1e-4 * (1 + abs(1 - math.sqrt(2)))
    cs = [O.Constraint(id_vec[None, :], [-1.0], O.ZeroSet(1)), O.Constraint(np.eye(n), np.zeros(n), O.ComplexPsdConeTriangle(n))]
    A, b, cones = O.assemble(cs)
    ref = O.solve(np.zeros((n, n)), vec_c, A, b, cones, O.Settings(kkt_solver="cg", eps_abs=1e-5, eps_rel=1e-5))
    assert ref.status == "Solved" and abs(res.iter - ref.iter) <= 25 and abs(res.obj_val - ref.obj_val) < 1e-5


def _vec_h(H):
    x = np.zeros(H.shape[0] ** 2); O.extract_upper_triangle_complex(H, x); return x


def _complex_infeasible_instances(r=3, seed=0):
    rng = np.random.default_rng(seed)
    G = rng.normal(size=(r, r)) + 1j * rng.normal(size=(r, r)); H1 = G @ G.conj().T        # Hermitian positive definite
    # primal infeasible: x >= 0 and  -I - x H1  Hermitian PSD  (internal form A x + s = b)
    A1 = sp.csc_matrix(np.concatenate([[-1.0], _vec_h(H1)]).reshape(-1, 1)); b1 = np.concatenate([[0.0], _vec_h(-np.eye(r))])
    # dual infeasible: min -x  s.t.  x H1 Hermitian PSD  (unbounded)
    A2 = sp.csc_matrix((-_vec_h(H1)).reshape(-1, 1)); b2 = np.zeros(r * r)
    return [("Primal_infeasible", np.array([1.0]), A1, b1, [("nonneg", 1), ("cplx", r * r)]),
            ("Dual_infeasible", np.array([-1.0]), A2, b2, [("cplx", r * r)])]


@pytest.mark.parametrize("r", [3, 6])
def test_complex_cone_infeasibility_certificates_match_oracle(r):
    """in_dual! / in_pol_recc! of the Hermitian cone = does cholesky!(Hermitian(X) + tol I) succeed (src/convexset.jl:415-424,
    src/algebra.jl:226-233); the device asks the same of the real embedding (one-workgroup Cholesky, csrc/psd_polar.hip)."""
    for want, q, A, b, kinds in _complex_infeasible_instances(r, seed=r):
        ocones = [O.Nonnegatives(d) if k == "nonneg" else O.ComplexPsdConeTriangle(d) for k, d in kinds]
        ref = O.solve(sp.csc_matrix((1, 1)), q, A, b, ocones, O.Settings(kkt_solver="cg", tol_constant=1e-10, tol_exponent=0.0))
        assert ref.status == want
        sets = [cj.Nonnegatives(d) if k == "nonneg" else cj.ComplexPsdConeTriangle(d) for k, d in kinds]
        md = cj.Model()
        md.set(sp.csc_matrix((1, 1)), q, A, b, sets, cj.Settings(kkt_solver=cj.with_options(cj.CGIndirectKKTSolver, tol_constant=1e-10, tol_exponent=0.0)))
        res = cj.optimize(md)
        assert res.status == want
        assert abs(res.iter - ref.iter) <= 40
