"""GPU parity tests for the PSD cone projections through the C ABI, against the oracle's LAPACK
dsyevr + syrk restatement of src/convexset.jl:219-263.  Tolerance: ||dX+||_F <= 64 d eps ||X||_F ; rank
(nnz_lambda) equal when the spectrum has a gap at 0."""
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


def _handle_for_sets(sets, dtype=np.float64):
    m = sum(K.dim for K in sets)
    h = cj.Handle(0, dtype=dtype)
    h.set_problem(sp.identity(2, format="csc"), np.zeros(2), sp.csc_matrix((m, 2)), np.zeros(m))
    h.set_cones([K.kind for K in sets], [K.dim for K in sets], None, None)
    return h


def sym_with_spectrum(rng, lam):
    d = lam.size
    Q = np.linalg.qr(rng.standard_normal((d, d)))[0]
    X = (Q * lam) @ Q.T
    return (X + X.T) / 2


def gapped_spectrum(rng, d, npos=None):
    npos = rng.integers(0, d + 1) if npos is None else npos
    lam = np.concatenate([rng.uniform(0.1, 2.0, npos), -rng.uniform(0.1, 2.0, d - npos)])
    rng.shuffle(lam)
    return lam


def check_projection(sets, mats, rtol_factor=64.0):
    h = _handle_for_sets(sets)
    s = []
    for K, X in zip(sets, mats):
        s.append(cj.problems.svec(X) if K.kind == F.PSD_TRIANGLE else X.reshape(-1, order="F"))
    s = np.concatenate(s)
    ref = s.copy(); info = {}
    O.project(ref, util.oracle_cones(sets), info)
    out, ranks, _ = h.project(s)
    off = 0
    for K, X, rk_ref, rk in zip(sets, mats, info["psd_rank"], ranks):
        d = X.shape[0]
        a = out[off:off + K.dim]; b = ref[off:off + K.dim]
        if K.kind == F.PSD_SQUARE:
            A = a.reshape(d, d, order="F")
            assert np.array_equal(A, A.T)                      # lower triangle is an exact mirror (convexset.jl:316-318)
            err = np.linalg.norm(a - b)
        else:
            err = np.linalg.norm(a - b)                        # svec is an isometry: this is the Frobenius norm
        assert err <= rtol_factor * d * EPS * max(np.linalg.norm(X), 1e-300), (d, err, err / (d * EPS * np.linalg.norm(X)))
        assert rk == rk_ref, (d, rk, rk_ref)
        off += K.dim
    return out


@pytest.mark.parametrize("kind", ["tri", "square"])
def test_psd_tiny_sizes(kind):
    rng = np.random.default_rng(1)
    dims = [2, 3, 4, 5, 7, 8, 9, 15, 16] * 3
    mats = [sym_with_spectrum(rng, gapped_spectrum(rng, d)) for d in dims]
    sets = [cj.PsdConeTriangle(d * (d + 1) // 2) if kind == "tri" else cj.PsdCone(d * d) for d in dims]
    check_projection(sets, mats)


@pytest.mark.parametrize("path", ["sign", "jacobi"])
@pytest.mark.parametrize("kind", ["tri", "square"])
def test_psd_workgroup_sizes(kind, path, monkeypatch):
    # 16 < d <= 256: batched matrix-sign iteration by default (csrc/psd.hip: polar_min = 16); the one-workgroup block-Jacobi
    # eigensolvers (still used by the definiteness tests of the infeasibility certificates) through COSMO_HIP_POLAR_BATCH_MIN=256
    monkeypatch.setenv("COSMO_HIP_POLAR_BATCH_MIN", "16" if path == "sign" else "256")
    rng = np.random.default_rng(2)
    dims = [17, 20, 24, 31, 32, 33, 47, 64, 65, 100, 127, 128, 129, 200, 255, 256]
    mats = [sym_with_spectrum(rng, gapped_spectrum(rng, d)) for d in dims]
    sets = [cj.PsdConeTriangle(d * (d + 1) // 2) if kind == "tri" else cj.PsdCone(d * d) for d in dims]
    check_projection(sets, mats)


def test_psd_large_multi_workgroup():
    rng = np.random.default_rng(3)
    dims = [257, 300, 520]
    mats = [sym_with_spectrum(rng, gapped_spectrum(rng, d)) for d in dims]
    sets = [cj.PsdConeTriangle(d * (d + 1) // 2) for d in dims]
    check_projection(sets, mats)


@pytest.mark.parametrize("path", ["sign", "jacobi"])
def test_psd_special_spectra(path, monkeypatch):
    monkeypatch.setenv("COSMO_HIP_POLAR_BATCH_MIN", "16" if path == "sign" else "256")
    rng = np.random.default_rng(4)
    mats, sets = [], []
    for d in (6, 40, 130):
        B = rng.standard_normal((d, d))
        mats += [B @ B.T / d + 0.2 * np.eye(d),                 # already PSD: projection is the identity map
                 -(B @ B.T / d + 0.2 * np.eye(d)),              # negative definite: projection is 0
                 np.zeros((d, d)),                              # zero matrix
                 3.0 * np.eye(d),                               # one eigenvalue of multiplicity d
                 sym_with_spectrum(rng, np.concatenate([np.full(d // 2, 1.5), np.full(d - d // 2, -0.7)])),  # two clusters
                 sym_with_spectrum(rng, np.concatenate([[5.0, -5.0], gapped_spectrum(rng, d - 2)])),        # +/- pair
                 1e8 * sym_with_spectrum(rng, gapped_spectrum(rng, d)),   # badly scaled
                 1e-9 * sym_with_spectrum(rng, gapped_spectrum(rng, d))]
        sets += [cj.PsdConeTriangle(d * (d + 1) // 2)] * 8
    out = check_projection(sets, mats)
    # exact zeros for the zero matrix
    off = 0
    for K, X in zip(sets, mats):
        if not X.any():
            assert not out[off:off + K.dim].any()
        off += K.dim


def test_psd_one_by_one_and_mixed_composite():
    rng = np.random.default_rng(5)
    sets = [cj.Nonnegatives(5), cj.PsdConeTriangle(1), cj.PsdCone(1), cj.SecondOrderCone(4), cj.PsdConeTriangle(10),
            cj.PsdCone(9), cj.PsdConeTriangle(210), cj.ZeroSet(2)]
    m = sum(K.dim for K in sets)
    h = _handle_for_sets(sets)
    s = rng.standard_normal(m)
    ref = s.copy(); info = {}
    O.project(ref, util.oracle_cones(sets), info)
    out, ranks, br = h.project(s)
    assert np.linalg.norm(out - ref) <= 64 * 20 * EPS * np.linalg.norm(s)
    assert [r for r in ranks if r >= 0] == info["psd_rank"]
    assert [b for b in br if b >= 0] == info["soc_branch"]


def test_psd_idempotent_and_in_cone_random_unstructured():
    # no spectral gap here: only properties (sets.jl:73-78: min eig >= -1e-9) and idempotence
    rng = np.random.default_rng(6)
    dims = [12, 50, 90, 180]
    mats = [(lambda B: (B + B.T) / 2)(rng.standard_normal((d, d))) for d in dims]
    sets = [cj.PsdConeTriangle(d * (d + 1) // 2) for d in dims]
    h = _handle_for_sets(sets)
    s = np.concatenate([cj.problems.svec(X) for X in mats])
    p1, r1, _ = h.project(s)
    p2, r2, _ = h.project(p1)
    off = 0
    for K, X in zip(sets, mats):
        d = X.shape[0]
        Xp = cj.problems.smat(p1[off:off + K.dim])
        assert np.linalg.eigvalsh(Xp).min() >= -1e-9
        assert np.linalg.norm(p2[off:off + K.dim] - p1[off:off + K.dim]) <= 64 * d * EPS * np.linalg.norm(X)
        w, V = np.linalg.eigh(X)
        assert np.linalg.norm(Xp - (V * np.maximum(w, 0)) @ V.T) <= 64 * d * EPS * np.linalg.norm(X)
        off += K.dim


def test_closest_correlation_end_to_end_small():
    # structure of test/UnitTests/closestcorr.jl:41-76 with the triangle cone, d = 30
    prob = cj.problems.closest_correlation(d=30, seed=12345)
    d = 30
    st = cj.Settings(eps_abs=1e-4, eps_rel=1e-4)
    model = cj.Model(); model.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], st)
    res = cj.optimize(model)
    X = cj.problems.smat(res.x)
    assert res.status == "Solved"                                              # closestcorr.jl:74
    assert np.max(np.abs(np.diag(X) - 1)) < 1e-5                               # :75
    assert np.linalg.eigvalsh(X).min() > -1e-3                                 # :76
    ref = O.solve(prob["P"], prob["q"], prob["A"], prob["b"], util.oracle_cones(prob["sets"]),
                  O.Settings(kkt_solver="cg", eps_abs=1e-4, eps_rel=1e-4))
    assert ref.status == "Solved" and abs(res.iter - ref.iter) <= 25
    assert abs(res.obj_val - ref.obj_val) <= 1e-4 * (1 + abs(ref.obj_val))
    assert np.max(np.abs(res.x - ref.x)) <= 1e-3


def test_chordal_sdp_end_to_end_small():
    prob = cj.problems.chordal_sdp(ncliques=12, dmin=4, dmax=40, sep_min=1, sep_max=3, n_total=1500, n_zero=10, n_nonneg=20)
    model = cj.Model(); model.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], cj.Settings())
    res = cj.optimize(model)
    ref = O.solve(prob["P"], prob["q"], prob["A"], prob["b"], util.oracle_cones(prob["sets"]), O.Settings(kkt_solver="cg"))
    assert res.status == ref.status == "Solved"
    assert abs(res.iter - ref.iter) <= 25
    assert abs(res.obj_val - ref.obj_val) <= 1e-4 * (1 + abs(ref.obj_val))
    assert len(res.info.rho_updates) == len(ref.rho_updates)


# ---------------------------------------------------------------------------------------------------------------------
# d > 256: matrix-sign (polar) iteration on the matrix cores (csrc/psd_polar.hip)
# ---------------------------------------------------------------------------------------------------------------------
def test_psd_polar_special_spectra_and_square_layout():
    rng = np.random.default_rng(14)
    d = 300
    B = rng.standard_normal((d, d))
    mats = [B @ B.T / d + 0.2 * np.eye(d),                      # PSD: identity map
            -(B @ B.T / d + 0.2 * np.eye(d)),                   # negative definite: 0
            np.zeros((d, d)),
            3.0 * np.eye(d),
            sym_with_spectrum(rng, np.concatenate([np.full(d // 2, 1.5), np.full(d - d // 2, -0.7)])),
            1e8 * sym_with_spectrum(rng, gapped_spectrum(rng, d)),
            1e-9 * sym_with_spectrum(rng, gapped_spectrum(rng, d))]
    sets = [cj.PsdConeTriangle(d * (d + 1) // 2)] * len(mats)
    out = check_projection(sets, mats, rtol_factor=8.0)        # the sign iteration is far more accurate than the 64 d eps bound
    assert not out[2 * sets[0].dim:3 * sets[0].dim].any()      # exact zeros for the zero matrix
    # square layout incl. an unsymmetric input (symmetrize_upper!, src/algebra.jl:201-208) and the exact mirror
    d = 280
    X = sym_with_spectrum(rng, gapped_spectrum(rng, d))
    N = rng.standard_normal((d, d)) * 1e-3
    h = _handle_for_sets([cj.PsdCone(d * d)])
    s = (X + N).reshape(-1, order="F")
    ref = s.copy(); O.project(ref, util.oracle_cones([cj.PsdCone(d * d)]))
    out, rk, _ = h.project(s)
    A = out.reshape(d, d, order="F")
    assert np.array_equal(A, A.T)
    assert np.linalg.norm(out - ref) <= 8 * d * EPS * np.linalg.norm(X)


def test_psd_polar_spectrum_with_tiny_eigenvalues():
    """Eigenvalues spread over 1e-14 .. 1 (the spectra ADMM iterates of low-rank SDPs have): eigenvalues below the cut-off
    ~3e-12 ||X||_F keep |sign| < 1, which perturbs X+ by less than their own size -- the error bound still holds; the rank
    (a by-product, trace of the sign matrix) is then only approximate."""
    rng = np.random.default_rng(15)
    d = 400
    lam = np.concatenate([rng.uniform(0.5, 2, 30), 10.0 ** rng.uniform(-14, -6, d - 60) * rng.choice([-1, 1], d - 60), -rng.uniform(0.1, 3, 30)])
    X = sym_with_spectrum(rng, lam)
    K = cj.PsdConeTriangle(d * (d + 1) // 2)
    h = _handle_for_sets([K])
    s = cj.problems.svec(X)
    ref = s.copy(); info = {}
    O.project(ref, util.oracle_cones([K]), info)
    out, rk, _ = h.project(s)
    assert np.linalg.norm(out - ref) <= 64 * d * EPS * np.linalg.norm(X)
    assert abs(int(rk[0]) - info["psd_rank"][0]) <= d - 60      # approximate by design
    # idempotent to rounding and inside the cone
    out2, _, _ = h.project(out)
    assert np.linalg.norm(out2 - out) <= 64 * d * EPS * np.linalg.norm(X)
    assert np.linalg.eigvalsh(cj.problems.smat(out)).min() >= -64 * d * EPS * np.linalg.norm(X)


RAGGED_SIDES = [17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 80, 81, 96, 100, 112, 127, 128, 129, 130, 144, 145, 160, 176, 177, 191, 192, 193, 200, 208,
                209, 224, 240, 241, 255, 256]


@pytest.mark.parametrize("occ", ["4", "3"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_batched_sign_path_ragged_tiles_are_bit_identical_to_the_quadrant_kernel(dtype, occ, monkeypatch):
    """The block-balanced ragged product kernel (k_symm_gemm_batch_r: sides rounded to 16, 1-4 blocks per part, upper blocks only on
    the diagonal, blocks dealt to the four waves) issues the same matrix instruction in the same k order per output element as the
    64 x 64 quadrant kernel it replaces: every projected cone is bit-identical, for every way a side can sit in the 16 / 64 grid -- and
    both are within the 64 d eps bound of dsyevr (src/convexset.jl:219-263)."""
    rng = np.random.default_rng(2030)
    mats = [sym_with_spectrum(rng, gapped_spectrum(rng, d)) * rng.uniform(0.1, 10.0) for d in RAGGED_SIDES]
    sets = [cj.PsdConeTriangle(d * (d + 1) // 2) for d in RAGGED_SIDES]
    s = np.concatenate([cj.problems.svec(X) for X in mats]).astype(dtype)
    # poison the allocator first: a handle of the same shape projects NaNs (its work matrices end up full of NaN) and is destroyed, so that the
    # next handle's work matrices start from that memory -- the ragged kernel writes only the d16 x d16 corner of a cone's buffers and the
    # verification sums run over the whole padded buffers, which therefore must be zero-initialised by the plan (a round-3 bug: every Float32
    # projection of a chordal SDP failed its verification with a NaN error bound)
    hp = _handle_for_sets(sets, dtype)
    hp.project(np.full(s.size, np.nan, dtype=dtype))
    hp.close()
    outs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("COSMO_HIP_POLAR_BATCH_RAGGED", flag)
        if flag == "1":
            monkeypatch.setenv("COSMO_HIP_POLAR_BATCH_OCC", occ)     # both register-allocation variants of the ragged kernel (4 is the default)
        else:
            monkeypatch.delenv("COSMO_HIP_POLAR_BATCH_OCC", raising=False)
        h = _handle_for_sets(sets, dtype)
        out, ranks, _ = h.project(s)
        ps = h.polar_stats()
        assert ps["batch_cones"] == len(sets) and ps["fallback_rounds"] == 0 and ps["unverified"] == 0
        outs[flag] = (out, np.asarray(ranks), h.time_psd_product(1, 3)[1])
        h.close()
    assert np.array_equal(outs["0"][0], outs["1"][0]) and np.array_equal(outs["0"][1], outs["1"][1])
    assert outs["1"][2] < 0.75 * outs["0"][2]                  # matrix-instruction flops actually issued per product: ragged < quadrant kernel
    useful = sum(2.0 * d * d * d for d in RAGGED_SIDES)       # full (unsymmetric) product of the d x d operands
    assert outs["1"][2] <= 0.5 * 1.4 * useful * 1.15           # upper blocks only: ~half of the full product, <= 1.4x padding (+ rounding of small sides)
    if dtype == np.float64:
        check_projection(sets, mats)                           # and the ragged kernel (default) against LAPACK


def test_closest_correlation_end_to_end_polar_path():
    d = 288
    prob = cj.problems.closest_correlation(d=d, seed=7)
    st = dict(max_iter=60, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9)
    model = cj.Model(); model.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], cj.Settings(**st))
    res = cj.optimize(model)
    ref = O.solve(prob["P"], prob["q"], prob["A"], prob["b"], util.oracle_cones(prob["sets"]), O.Settings(kkt_solver="cg", **st))
    assert res.status == ref.status == "Max_iter_reached" and res.iter == ref.iter == 60
    assert np.max(np.abs(res.x - ref.x)) <= 1e-7 * max(1.0, np.max(np.abs(ref.x)))
    assert np.allclose(res.info.rho_updates, ref.rho_updates, rtol=1e-6)
    assert abs(res.info.r_prim - ref.r_prim) <= 1e-6 * max(ref.r_prim, 1e-12) + 1e-12


def test_cg_operator_split_is_the_same_operator(monkeypatch):
    """Singleton rows of A as a diagonal of A' rho A (csrc/api.hip: build_op_split): the split and the unsplit CG operator give the
    same KKT solutions, iteration counts and ADMM trajectories (a re-association, not a different algorithm)."""
    prob = cj.problems.chordal_sdp(ncliques=12, dmin=4, dmax=40, sep_min=1, sep_max=3, n_total=1500, n_zero=10, n_nonneg=20)
    out = {}
    for split in ("1", "0"):
        monkeypatch.setenv("COSMO_HIP_OP_SPLIT", split)
        model = cj.Model()
        model.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], cj.Settings(max_iter=60, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9,
                                                                                     kkt_solver=cj.with_options(cj.CGIndirectKKTSolver, tol_constant=1e-10, tol_exponent=0.0)))
        res = cj.optimize(model)
        rhs = np.random.default_rng(1).standard_normal(model.n + model.m)
        sol, its = model.handle.kkt_solve(rhs)
        out[split] = (res, sol, its)
    (r1, s1, k1), (r0, s0, k0) = out["1"], out["0"]
    assert r1.iter == r0.iter == 60 and abs(k1 - k0) <= 3               # a 1e-10 threshold on a ~140-iteration solve
    assert abs(r1.kkt_iters_total - r0.kkt_iters_total) <= 0.02 * r0.kkt_iters_total
    assert np.linalg.norm(s1 - s0) <= 1e-8 * np.linalg.norm(s0)
    assert np.max(np.abs(r1.x - r0.x)) <= 1e-7 * max(1.0, np.max(np.abs(r0.x)))
    assert np.allclose(r1.info.rho_updates, r0.info.rho_updates, rtol=1e-9)


# ---------------------------------------------------------------------------------------------------------------------
# per-cone lifting depth of the matrix-sign projections (round 4, csrc/psd_polar.hip PolarPlan::adapt): the number of lifting steps of every
# cone follows that cone's own verification history; the a-posteriori check -- and with it the error bound of EVERY projection -- is unchanged
# ---------------------------------------------------------------------------------------------------------------------
def test_adaptive_lifting_depth_keeps_every_projection_within_the_bound_and_saves_products(monkeypatch):
    """A batch of cones with very different spectra projected 40 times: (a) cones whose smallest |lambda| is large settle at a small depth,
    a cone with an eigenvalue of 1e-6 ||X|| stays deep; (b) EVERY one of the 40 projections of EVERY cone is within 64 d eps ||X||_F of the
    LAPACK projection (failed verifications take their fallback round: unverified == 0); (c) the d^3-weighted product count falls well below the
    fixed schedule's 44; (d) without the opt-in (the default) the schedule is the fixed one of round 3."""
    monkeypatch.setenv("COSMO_HIP_POLAR_ADAPT", "1")                    # opt-in (measured: fewer products, slower steps on config 5 -- see PolarPlan::adapt)
    rng = np.random.default_rng(404)
    dims = [24, 40, 64, 90, 130, 200, 33, 57]
    sets = [cj.PsdConeTriangle(d * (d + 1) // 2) for d in dims]
    mats = []
    for i, d in enumerate(dims):
        lam = gapped_spectrum(rng, d)
        if i == 3:
            lam[0] = 1e-6 * np.linalg.norm(lam)                       # needs ~9 lifting steps: must not be dragged down by its neighbours
        mats.append(sym_with_spectrum(rng, lam))
    s0 = np.concatenate([cj.problems.svec(M) for M in mats])
    ref = s0.copy()
    O.project(ref, util.oracle_cones(sets), {})
    offs = np.concatenate([[0], np.cumsum([K.dim for K in sets])])
    h = _handle_for_sets(sets)
    first = None
    for it in range(40):
        out, ranks, _ = h.project(s0 * (1.0 + 0.01 * it))              # the same spectra at a slowly changing scale, as consecutive ADMM iterates are
        for k, d in enumerate(dims):
            err = np.linalg.norm(out[offs[k]:offs[k + 1]] / (1.0 + 0.01 * it) - ref[offs[k]:offs[k + 1]])
            assert err <= 64 * d * EPS * np.linalg.norm(mats[k]), (it, k, err / (d * EPS * np.linalg.norm(mats[k])))
        st = h.polar_depth_stats()
        if it == 0:
            first = st
    ps = h.polar_stats()
    assert first["adaptive"] == 1 and first["depth_min"] == first["depth_max"] == 9 and abs(first["weighted_products_per_projection"] - 44.0) < 1e-9
    assert ps["unverified"] == 0 and ps["batch_cones"] == len(sets)
    assert st["depth_min"] <= 4 and st["depth_max"] >= 8, st            # well-separated spectra went down, the 1e-6 cone did not
    assert st["weighted_products_per_projection"] <= 38.0 and st["downward_probes"] > 0, st
    assert st["failed_verifications"] >= 1                              # the probing really found the edge (and the fallback round repaired it)
    h.close()
    monkeypatch.setenv("COSMO_HIP_POLAR_ADAPT", "0")
    h = _handle_for_sets(sets)
    for it in range(8):
        out0, _, _ = h.project(s0)
    st0 = h.polar_depth_stats()
    assert st0["adaptive"] == 0 and st0["depth_min"] == st0["depth_max"] == 9 and abs(st0["weighted_products_per_projection"] - 44.0) < 1e-9
    assert h.polar_stats()["fallback_rounds"] == 0
    h.close()


def test_adaptive_lifting_depth_large_cone_and_loop_level_agreement(monkeypatch):
    """(a) A single large cone (d = 320 > 256: the per-cone path) projected repeatedly: its depth goes down, every projection within the bound.
    (b) Loop level: a small chordal SDP solved with and without the adaptive depth -- same status, iteration count within one check interval,
    objective to 1e-6 (the projections differ by at most their verified bound ~1e-13 ||X||_F per iteration)."""
    monkeypatch.setenv("COSMO_HIP_POLAR_ADAPT", "1")
    rng = np.random.default_rng(77)
    d = 320
    K = cj.PsdConeTriangle(d * (d + 1) // 2)
    M = sym_with_spectrum(rng, gapped_spectrum(rng, d))
    s0 = cj.problems.svec(M)
    ref = s0.copy(); O.project(ref, util.oracle_cones([K]), {})
    h = _handle_for_sets([K])
    prods = []
    for it in range(24):
        out, _, _ = h.project(s0)
        assert np.linalg.norm(out - ref) <= 64 * d * EPS * np.linalg.norm(M)
        prods.append(h.polar_stats()["products_last_large"])
    assert h.polar_stats()["large_cones"] == 1 and h.polar_stats()["unverified"] == 0
    assert prods[0] > prods[-1] and h.polar_depth_stats()["downward_probes"] > 0, prods
    h.close()
    prob = cj.problems.chordal_sdp(ncliques=12, dmin=18, dmax=70, sep_min=1, sep_max=3, n_total=2500, n_zero=40, n_nonneg=80)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("COSMO_HIP_POLAR_ADAPT", flag)
        md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], cj.Settings(kkt_solver=cj.CGIndirectKKTSolver, max_iter=3000))
        res[flag] = cj.optimize(md)
    a, b = res["1"], res["0"]
    assert a.status == b.status == "Solved" and abs(a.iter - b.iter) <= 25 and abs(a.obj_val - b.obj_val) <= 1e-6 * (1 + abs(b.obj_val))


# ---- cosmo_hip_set_psd_projection(EIGEN): the eigendecomposition-based projection as a first-class option (round 6; VERDICT r05 "missing" 5) ----------
def _project_with_mode(sets, mats, mode):
    h = _handle_for_sets(sets)
    h.set_psd_projection(mode)
    s = np.concatenate([cj.problems.svec(X) if K.kind == F.PSD_TRIANGLE else X.reshape(-1, order="F") for K, X in zip(sets, mats)])
    out, ranks, _ = h.project(s)
    return out, list(ranks)


def test_eigen_mode_counts_nnz_lambda_from_the_eigenvalues():
    """north_star's wording of a7 / a8 is the reference's eigendecomposition-based projection (src/convexset.jl:163-189, 243-263: syevr! for the
    eigenpairs above zero, nnz_lambda = their number, rank-k update).  EIGEN mode runs that shape at every side on the Jacobi eigensolvers: on spectra
    WITHOUT a gap at zero -- eigenvalues of both signs at 1e-9 .. 1e-13 of ||X|| next to O(1) ones, where the sign iteration's trace count is
    'approximate by design' -- nnz_lambda equals LAPACK's count for every eigenvalue the eigensolver resolves (|lambda| >= 1e-10 ||X||_F here), and the
    projection stays within the 64 d eps bound.  Sides 12 (wave kernel), 40 / 130 / 250 (one workgroup), 300 (multi-workgroup, host-paced sweeps)."""
    rng = np.random.default_rng(41)
    sets, mats, want = [], [], []
    for d in (12, 40, 130, 250, 300):
        npos_big, nneg_big = d // 3, d // 3
        small = d - npos_big - nneg_big
        sp_small = np.concatenate([+10.0 ** rng.uniform(-9.5, -7.0, small // 2), -10.0 ** rng.uniform(-9.5, -7.0, small - small // 2)])
        lam = np.concatenate([rng.uniform(0.5, 2.0, npos_big), -rng.uniform(0.5, 2.0, nneg_big), sp_small])
        rng.shuffle(lam)
        X = sym_with_spectrum(rng, lam)
        mats.append(X); sets.append(cj.PsdConeTriangle(d * (d + 1) // 2))
        want.append(int(np.sum(np.linalg.eigvalsh(X) > 0)))
    out_e, rk_e = _project_with_mode(sets, mats, F.PSD_PROJECTION_EIGEN)
    out_s, rk_s = _project_with_mode(sets, mats, F.PSD_PROJECTION_SIGN)
    off = 0
    for K, X, w, re_, rs_ in zip(sets, mats, want, rk_e, rk_s):
        d = X.shape[0]
        lam, V = np.linalg.eigh(X)
        ref = cj.problems.svec((V * np.maximum(lam, 0.0)) @ V.T)
        for out in (out_e, out_s):
            assert np.linalg.norm(out[off:off + K.dim] - ref) <= 64.0 * d * EPS * np.linalg.norm(X), d
        assert re_ == w, (d, re_, w)                           # EIGEN: the count of positive eigenvalues itself
        assert abs(rs_ - w) <= d - 2 * (d // 3)                 # SIGN: may put the near-zero cluster on either side (inside its bound)
        off += K.dim


def test_eigen_mode_through_the_loop_and_against_the_sign_path():
    """A closest-correlation SDP (d = 40) and a chordal SDP with cliques up to side 110, solved with psd_projection = 'eigen' and with the default:
    same status, same iteration count, objectives to 1e-6; the eigen run reports Jacobi sweeps (cosmo_hip_psd_stats), the default none above side 16."""
    for prob in (cj.problems.closest_correlation(d=40, seed=3),
                 cj.problems.chordal_sdp(ncliques=10, dmin=6, dmax=110, sep_min=1, sep_max=4, n_total=1500, n_zero=10, n_nonneg=20, seed=12)):
        res = {}
        for mode in ("sign", "eigen"):
            md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], cj.Settings(psd_projection=mode, eps_abs=1e-5, eps_rel=1e-5))
            res[mode] = cj.optimize(md)
        a, b = res["sign"], res["eigen"]
        assert a.status == b.status == "Solved" and abs(a.iter - b.iter) <= 25, (a.status, b.status, a.iter, b.iter)
        assert abs(a.obj_val - b.obj_val) <= 1e-6 * (1 + abs(a.obj_val))
    with pytest.raises(ValueError):
        md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], cj.Settings(psd_projection="qr")); cj.optimize(md)
