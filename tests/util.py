"""Shared helpers of the test-suite: model sets <-> oracle cones, scaled-problem setup through the oracle."""
import numpy as np
import scipy.sparse as sp

import cosmo_jl_amd as cj
from oracle import cosmo_oracle as O

EPS = np.finfo(np.float64).eps


def oracle_cones(sets):
    out = []
    for K in sets:
        if K.kind == cj._ffi.BOX:
            out.append(O.Box(K.l, K.u))
        elif K.kind in (cj._ffi.EXP, cj._ffi.DUAL_EXP):
            out.append(O.Cone(K.kind, 3, max_iter=100, tol=1e-8))
        elif K.kind in (cj._ffi.POW, cj._ffi.DUAL_POW):
            out.append(O.Cone(K.kind, 3, alpha=K.alpha, max_iter=20, tol=1e-8))
        else:
            out.append(O.Cone(K.kind, K.dim, constr_type=(np.zeros(K.dim, dtype=bool) if K.kind == cj._ffi.NONNEG else None)))
    return out


def model_sets(cones):
    m = {O.ZERO: cj.ZeroSet, O.NONNEG: cj.Nonnegatives, O.SOC: cj.SecondOrderCone, O.PSD_SQUARE: cj.PsdCone,
         O.PSD_TRIANGLE: cj.PsdConeTriangle, O.PSD_TRIANGLE_COMPLEX: cj.ComplexPsdConeTriangle}
    out = []
    for c in cones:
        if c.kind == O.BOX:
            out.append(cj.Box(c.l, c.u))
        elif c.kind in (O.EXP, O.DUAL_EXP):
            out.append((cj.ExponentialCone if c.kind == O.EXP else cj.DualExponentialCone)())
        elif c.kind in (O.POW, O.DUAL_POW):
            out.append((cj.PowerCone if c.kind == O.POW else cj.DualPowerCone)(c.alpha))
        else:
            out.append(m[c.kind](c.dim))
    return out


def oracle_settings(**kw):
    return O.Settings(**kw)


def make_handle_from_workspace(ws: O.Workspace, kkt_kind=cj._ffi.KKT_CG, dtype=np.float64, **param_overrides):
    """Feeds the oracle's ALREADY SCALED problem (what Julia's setup! would hand over) to the device library (dtype float32: rounded
    to Float32 at the boundary, libcosmo_hip_f32.so)."""
    h = cj.Handle(0, dtype=dtype)
    h.set_problem(ws.P, ws.q, ws.A, ws.b)
    bl = np.concatenate([c.l for c in ws.cones if c.kind == O.BOX] or [np.zeros(0)])
    bu = np.concatenate([c.u for c in ws.cones if c.kind == O.BOX] or [np.zeros(0)])
    h.set_cones([c.kind for c in ws.cones], [c.dim for c in ws.cones], bl, bu, cone_param=[c.alpha for c in ws.cones])
    st = ws.st
    p = h.default_params()
    p.kkt_kind = kkt_kind
    p.sigma, p.alpha, p.rho = st.sigma, st.alpha, st.rho
    p.eps_abs, p.eps_rel = st.eps_abs, st.eps_rel
    p.tol_constant, p.tol_exponent = st.tol_constant, st.tol_exponent
    p.rho_min, p.rho_max, p.rho_tol = st.RHO_MIN, st.RHO_MAX, st.RHO_TOL
    p.rho_eq_over_rho_ineq = st.RHO_EQ_OVER_RHO_INEQ
    p.adaptive_rho_tolerance = st.adaptive_rho_tolerance
    p.cosmo_infty_min_scaling = st.COSMO_INFTY * st.MIN_SCALING
    p.max_iter = st.max_iter
    p.adaptive_rho_max_adaptions = min(st.adaptive_rho_max_adaptions, 2 ** 62)
    p.check_termination = st.check_termination
    p.check_infeasibility = st.check_infeasibility
    p.adaptive_rho = 1 if st.adaptive_rho else 0
    p.adaptive_rho_interval = st.adaptive_rho_interval
    p.unscale_residuals = 1 if st.scaling != 0 else 0
    p.obj_true, p.obj_true_tol = st.obj_true, st.obj_true_tol
    for k, v in param_overrides.items():
        setattr(p, k, v)
    h.set_params(p)
    h.set_scaling_full(ws.sm.D, ws.sm.Dinv, ws.sm.E, ws.sm.Einv, ws.sm.c, ws.sm.cinv)
    return h


def random_qp(rng, n, m_zero, m_nonneg, m_box, soc_dims=(), density=0.1, psd_tri_dims=(), p_shift=0.1, psd_sq_dims=()):
    """A feasible random conic QP in internal form with a mix of cone types."""
    m_soc = int(sum(soc_dims))
    m_psd = int(sum(d * (d + 1) // 2 for d in psd_tri_dims)) + int(sum(d * d for d in psd_sq_dims))
    m = m_zero + m_nonneg + m_box + m_soc + m_psd
    A = sp.random(m, n, density=density, random_state=rng, format="csc", data_rvs=rng.standard_normal)
    A = (A + sp.csc_matrix((np.ones(min(m, n)), (np.arange(min(m, n)), np.arange(min(m, n)))), shape=(m, n)) * 0.5).tocsc()
    S = sp.random(n, n, density=min(1.0, 3.0 / n), random_state=rng, format="csc", data_rvs=rng.standard_normal)
    P = (S @ S.T + p_shift * sp.identity(n)).tocsc()
    q = rng.standard_normal(n)
    x0 = rng.standard_normal(n)
    s0 = [np.zeros(m_zero), rng.uniform(0.1, 1.0, m_nonneg)]
    sets = []
    if m_zero:
        sets.append(cj.ZeroSet(m_zero))
    if m_nonneg:
        sets.append(cj.Nonnegatives(m_nonneg))
    if m_box:
        sb = rng.standard_normal(m_box)
        l = sb - rng.uniform(0.1, 1.0, m_box); u = sb + rng.uniform(0.1, 1.0, m_box)
        k = max(1, m_box // 10)
        l[:k] = sb[:k]; u[:k] = sb[:k]                       # equality rows
        l[k:2 * k] = -1e30; u[k:2 * k] = 1e30                # loose rows
        l[2 * k:3 * k] = -np.inf                             # one-sided
        s0.append(sb)
        sets.append(cj.Box(l, u))
    for d in soc_dims:
        v = rng.standard_normal(d - 1)
        s0.append(np.concatenate([[np.linalg.norm(v) + 0.5], v]))
        sets.append(cj.SecondOrderCone(d))
    for d in psd_tri_dims:
        B = rng.standard_normal((d, d))
        s0.append(cj.problems.svec(B @ B.T / d + 0.1 * np.eye(d)))
        sets.append(cj.PsdConeTriangle(d * (d + 1) // 2))
    for d in psd_sq_dims:                                    # square PsdCone (vec layout), after the triangle cones
        B = rng.standard_normal((d, d))
        s0.append((B @ B.T / d + 0.1 * np.eye(d)).reshape(-1, order="F"))
        sets.append(cj.PsdCone(d * d))
    s0 = np.concatenate(s0)
    b = A @ x0 + s0
    return dict(P=P, q=q, A=A, b=b, sets=sets)


def nan_row_qp(seed, n, m, nan_row, split=None):
    """min 1/2 x'Px + q'x  s.t.  b - Ax >= 0 (Nonnegatives, strictly feasible at 0), row `nan_row` of A empty with b = NaN (None: clean)."""
    rng = np.random.default_rng(seed)
    A = sp.random(m, n, density=min(1.0, 6.0 / n), random_state=rng, format="lil", data_rvs=rng.standard_normal)
    b = rng.uniform(0.5, 2.0, m)
    if nan_row is not None:
        A[nan_row, :] = 0.0
    A = A.tocsc(); A.eliminate_zeros()
    if nan_row is not None:
        assert A.tocsr().indptr[nan_row] == A.tocsr().indptr[nan_row + 1]
        b[nan_row] = np.nan
    P = sp.identity(n, format="csc") * 2.0
    q = rng.standard_normal(n)
    dims = [m] if split is None else [split, m - split]
    return dict(P=P, q=q, A=A, b=b, sets=[cj.Nonnegatives(d) for d in dims])
