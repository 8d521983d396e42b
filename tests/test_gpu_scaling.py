"""GPU tests of the device Ruiz equilibration (cosmo_hip_scale_ruiz; scale_ruiz!, src/scaling.jl:21-116):
scaling matrices, scaled data, scaled Box bounds and rho classes against the oracle's restatement, and the end-to-end
solve with scaling on the device against scaling on the host."""
import numpy as np
import pytest
import scipy.sparse as sp

import cosmo_jl_amd as cj
from oracle import cosmo_oracle as O
from tests import util

pytestmark = pytest.mark.gpu
F = cj._ffi


def _mixed_problem(seed, n=60):
    rng = np.random.default_rng(seed)
    kinds = [F.ZERO, F.NONNEG, F.BOX, F.SOC, F.SOC, F.PSD_TRIANGLE, F.EXP, F.POW]
    dims = [5, 12, 20, 7, 4, 10, 3, 3]
    alphas = [0, 0, 0, 0, 0, 0, 0, 0.4]
    m = sum(dims)
    A = sp.random(m, n, density=0.2, random_state=seed, data_rvs=lambda k: rng.normal(size=k) * 10.0 ** rng.integers(-3, 4, size=k)).tocsc()
    A = (A + sp.csc_matrix((np.full(min(m, n), 0.5), (np.arange(min(m, n)), np.arange(min(m, n)))), shape=(m, n))).tocsc()
    S = sp.random(n, n, density=0.1, random_state=seed + 1).tocsc()
    P = (S + S.T + sp.identity(n) * rng.uniform(0.1, 30.0)).tocsc()
    P.sort_indices(); A.sort_indices()
    q = rng.normal(size=n) * 50
    b = rng.normal(size=m)
    l = -np.abs(rng.normal(size=20)); u = np.abs(rng.normal(size=20))
    l[3] = -np.inf; u[3] = np.inf; l[5] = u[5] = 0.25; u[7] = np.inf
    return P, q, A, b, kinds, dims, alphas, l, u


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_device_ruiz_matches_oracle(seed):
    P, q, A, b, kinds, dims, alphas, l, u = _mixed_problem(seed)
    n, m = P.shape[0], A.shape[0]
    # oracle
    cones = []
    for k, d, a in zip(kinds, dims, alphas):
        cones.append(O.Box(l, u) if k == O.BOX else O.Cone(k, d, alpha=a, constr_type=(np.zeros(d, dtype=bool) if k == O.NONNEG else None)))
    st = O.Settings()
    Po, qo, Ao, bo = P.copy(), q.copy(), A.copy(), b.copy()
    sm = O.scale_ruiz(Po, qo, Ao, bo, cones, st)
    O.classify_constraints(cones, bo, st)
    cls_ref = O.row_rho_class(cones)
    # device
    h = cj.Handle(0)
    h.set_problem(P, q, A, b)
    h.set_cones(kinds, dims, l, u, cone_param=alphas)
    D, E, c = h.scale_ruiz(st.scaling, st.MIN_SCALING, st.MAX_SCALING)
    assert np.array_equal(D, sm.D)                       # inf-norms, sqrt, products: bit-exact
    assert np.allclose(E, sm.E, rtol=4e-16, atol=0)      # the per-cone mean (rectify_scalar_scaling!) is a sum
    assert abs(c - sm.c) <= 4e-16 * abs(sm.c)            # mean(col norms of P) is a sum
    assert np.array_equal(h.get_rho_classes(), cls_ref)
    # scaled operators: compare through SpMVs with random vectors
    rng = np.random.default_rng(seed + 10)
    x = rng.normal(size=n); y = rng.normal(size=m)
    assert np.allclose(h.spmv(F.MAT_A, x), Ao @ x, rtol=1e-13, atol=1e-13)
    assert np.allclose(h.spmv(F.MAT_AT, y), Ao.T @ y, rtol=1e-13, atol=1e-13)
    assert np.allclose(h.spmv(F.MAT_P, x), Po @ x, rtol=1e-13, atol=1e-13)
    # the merged operator [P | A'] must carry the same scaling as its parts: the same (possibly unconverged) CG solve on a
    # second handle that was fed the oracle-scaled data must walk the same iterates
    p = h.default_params(); p.kkt_kind = F.KKT_CG
    h.set_params(p)
    h2 = cj.Handle(0)
    h2.set_problem(Po, qo, Ao, bo)
    lo = np.concatenate([cn.l for cn in cones if cn.kind == O.BOX]); uo = np.concatenate([cn.u for cn in cones if cn.kind == O.BOX])
    h2.set_cones(kinds, dims, lo, uo, cone_param=alphas)
    h2.set_params(p)
    assert np.array_equal(h.get_rho_vec(), h2.get_rho_vec())
    rhs = rng.normal(size=n + m)
    sol, it1 = h.kkt_solve(rhs)
    sol2, it2 = h2.kkt_solve(rhs)
    assert it1 == it2
    assert np.linalg.norm(sol - sol2) <= 1e-9 * np.linalg.norm(sol2)
    h2.close()
    h.close()


def test_device_ruiz_rejects_bad_calls():
    P, q, A, b, kinds, dims, alphas, l, u = _mixed_problem(4)
    h = cj.Handle(0)
    h.set_problem(P, q, A, b)
    with pytest.raises(cj.CosmoHipError):
        h.scale_ruiz(10)                                  # cones missing
    h.set_cones(kinds, dims, l, u, cone_param=alphas)
    h.scale_ruiz(10)
    with pytest.raises(cj.CosmoHipError):
        h.scale_ruiz(10)                                  # already scaled
    Pn = P.tolil(); Pn[0, 1] += 1.0; Pn = Pn.tocsc()
    h.set_problem(Pn, q, A, b); h.set_cones(kinds, dims, l, u, cone_param=alphas)
    with pytest.raises(cj.CosmoHipError):
        h.scale_ruiz(10)                                  # unsymmetric P
    h.close()


@pytest.mark.parametrize("case", ["box_qp", "socp", "sdp"])
def test_solve_with_device_scaling_matches_host_scaling(case):
    if case == "box_qp":
        pr = cj.problems.sparse_box_qp(n=300, m=600, nnz=6000, seed=3)
    elif case == "socp":
        pr = cj.problems.socp(seed=1000)
    else:
        pr = cj.problems.closest_correlation(d=12, seed=4)
    res = []
    for dev in (True, False):
        model = cj.Model()
        model.set(pr["P"], pr["q"], pr["A"], pr["b"], pr["sets"], cj.Settings(device_scaling=dev, eps_abs=1e-6, eps_rel=1e-6))
        res.append(cj.optimize(model))
    a, b_ = res
    assert a.status == b_.status == "Solved"
    assert abs(a.iter - b_.iter) <= 25
    assert abs(a.obj_val - b_.obj_val) <= 1e-6 * max(1.0, abs(b_.obj_val))
    assert np.linalg.norm(a.x - b_.x) <= 1e-5 * max(1.0, np.linalg.norm(b_.x))


def test_cfg2_full_size_device_scaling_equals_host_scaling():
    """BASELINE config 2 size: the device equilibration reproduces the host restatement bit for bit (Box cone only => no
    rectification sums), and the loop started from either gives the same iterates."""
    prob = cj.problems.sparse_box_qp()
    n, m = prob["A"].shape[1], prob["A"].shape[0]
    out = []
    for dev in (True, False):
        model = cj.Model()
        model.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], cj.Settings(max_iter=30, eps_abs=0, eps_rel=0, device_scaling=dev))
        res = cj.optimize(model)
        x = np.random.default_rng(0).standard_normal(n)
        out.append((model.sm, res, model.handle.spmv(F.MAT_A, x), model.handle.get_rho_classes(), model.sets[0]))
    (smd, rd, axd, cd, Kd), (smh, rh, axh, ch, Kh) = out
    assert np.array_equal(smd.D, smh.D) and np.array_equal(smd.E, smh.E)
    assert abs(smd.c - smh.c) <= 4e-16 * smh.c
    assert np.array_equal(cd, ch)
    assert np.array_equal(Kd.l, Kh.l) and np.array_equal(Kd.u, Kh.u)
    assert np.array_equal(axd.view(np.int64), axh.view(np.int64))          # identical scaled A on the device
    assert rd.iter == rh.iter == 30
    assert np.linalg.norm(rd.x - rh.x) <= 1e-9 * np.linalg.norm(rh.x)
