"""Definition-level anchor for the oracle's Anderson accelerator (COSMOAccelerators.jl "^0.1.0" is NOT vendored in
/root/reference, so `oracle.AndersonAccelerator` restates the published type-II method with QR-updated least squares and a
restarted memory from the call sites src/accelerator_interface.jl:58-130).  CPU only.

Type-II Anderson acceleration (Walker & Ni 2011, eq. 2.1-2.2, the formulation the package documents): with residuals
f_i = x_i - g_i, dF = [f_{i+1} - f_i], dG = [g_{i+1} - g_i] over the stored history,

        eta = argmin || f_k - dF eta ||_2 ,        g_acc = g_k - dG eta .

That least-squares problem has a unique solution for a full-rank history, so ANY correct implementation -- whatever its QR
bookkeeping -- must produce this candidate up to rounding.  The test replays random and contractive fixed-point sequences through the
oracle's update / accelerate pair and compares every accepted candidate with a dense `lstsq` on the explicitly stored history,
including the memory restart after `mem` columns and the minimum-memory rule.
"""
import numpy as np
import pytest

from oracle import cosmo_oracle as O


def replay(dim, steps, mem, seed, contractive):
    rng = np.random.default_rng(seed)
    aa = O.AndersonAccelerator(dim, mem=mem)
    B = rng.standard_normal((dim, dim)); B = 0.9 * B / np.linalg.norm(B, 2)
    c = rng.standard_normal(dim)
    x = rng.standard_normal(dim)
    hist_f, hist_g = [], []          # explicit history since the last restart (columns of dF / dG are differences of these)
    checked = 0
    for it in range(steps):
        g = (B @ x + c) if contractive else rng.standard_normal(dim)
        restarts_before = aa.num_restarts
        was_init = aa.init_phase
        aa.update(g, x)
        f = x - g
        if aa.num_restarts != restarts_before:                # RestartedMemory: the column just added is the first of a new history
            hist_f, hist_g = hist_f[-1:], hist_g[-1:]
        hist_f.append(f.copy()); hist_g.append(g.copy())
        if was_init:
            assert aa.iter == 0                               # the first pair only initialises x_last / g_last / f_last
        cand = g.copy()
        aa.accelerate(cand)
        ncol = len(hist_f) - 1
        assert min(aa.iter, aa.mem) == ncol
        if ncol < aa.min_mem:
            assert not aa.was_successful() and np.array_equal(cand, g)      # minimum-memory rule: no acceleration yet
        elif aa.was_successful():
            dF = np.column_stack([hist_f[i + 1] - hist_f[i] for i in range(ncol)])
            dG = np.column_stack([hist_g[i + 1] - hist_g[i] for i in range(ncol)])
            eta = np.linalg.lstsq(dF, f, rcond=None)[0]
            want = g - dG @ eta
            cond = np.linalg.cond(dF)
            assert np.linalg.norm(cand - want) <= 1e-11 * cond * max(1.0, np.linalg.norm(want)), (it, ncol, cond)
            # optimality of eta itself: the residual f - dF eta is orthogonal to the history
            r = f - dF @ aa.eta[:ncol]
            assert np.linalg.norm(dF.T @ r) <= 1e-10 * cond * np.linalg.norm(dF, 2) * max(np.linalg.norm(f), 1e-300)
            checked += 1
        x = cand if contractive else rng.standard_normal(dim)
    return checked, aa


@pytest.mark.parametrize("contractive", [False, True])
@pytest.mark.parametrize("dim,mem", [(40, 15), (12, 5), (200, 15)])
def test_accelerated_candidate_is_the_type2_least_squares_point(dim, mem, contractive):
    checked, aa = replay(dim, 3 * mem + 7, mem, seed=dim + mem, contractive=contractive)
    assert checked >= mem                                      # many accepted steps were compared
    assert aa.num_restarts >= 2                                # ... across at least two memory restarts


def test_anderson_on_a_linear_fixed_point_map_terminates_like_gmres():
    """Property of the definition (Walker & Ni, Theorem 2.2): on a linear contraction x -> B x + c, Anderson without truncation is
    essentially GMRES on (I - B) x = c, so with memory >= dim the fixed point is reached (to rounding) within dim + 1 accelerated
    steps."""
    dim = 8
    rng = np.random.default_rng(3)
    B = rng.standard_normal((dim, dim)); B = 0.5 * B / np.linalg.norm(B, 2)
    c = rng.standard_normal(dim)
    xs = np.linalg.solve(np.eye(dim) - B, c)
    aa = O.AndersonAccelerator(dim, mem=dim, min_mem=1)
    x = np.zeros(dim)
    errs = []
    for _ in range(dim + 2):
        g = B @ x + c
        aa.update(g, x)
        aa.accelerate(g)
        x = g
        errs.append(np.linalg.norm(x - xs))
    assert min(errs) <= 1e-9 * np.linalg.norm(xs)


# ---- the non-default variants (docs/src/acceleration.md:23-26): Type1 / Type2{NormalEquations}, RestartedMemory / RollingMemory ----------------
def replay_ne(dim, steps, mem, seed, type1, rolling):
    """Replays a contractive fixed-point sequence through oracle.AndersonAcceleratorNE and compares every accepted candidate with the DEFINITION on an
    explicitly stored history: with dX / dF / dG the differences of the last <= mem stored (x, f, g) triples (all of them since the last restart for a
    restarted memory, the most recent `mem` for a rolling one),
        Type1:  eta solves (dX' dF) eta = dX' f        Type2:  eta = argmin ||f - dF eta||_2        candidate = g - dG eta."""
    rng = np.random.default_rng(seed)
    aa = O.AndersonAcceleratorNE(dim, mem=mem, type1=type1, rolling=rolling)
    B = rng.standard_normal((dim, dim)); B = 0.9 * B / np.linalg.norm(B, 2)
    c = rng.standard_normal(dim)
    x = rng.standard_normal(dim)
    hist = []                                            # (x, f, g) since the last restart
    checked = 0
    for it in range(steps):
        g = B @ x + c
        restarts_before = aa.num_restarts
        aa.update(g, x)
        f = x - g
        if aa.num_restarts != restarts_before:
            hist = hist[-1:]
        hist.append((x.copy(), f.copy(), g.copy()))
        cand = g.copy()
        aa.accelerate(cand)
        ncol_all = len(hist) - 1
        ncol = min(ncol_all, aa.mem)
        assert min(aa.iter, aa.mem) == ncol
        if ncol < aa.min_mem:
            assert not aa.was_successful() and np.array_equal(cand, g)
        elif aa.was_successful():
            cols = range(ncol_all - ncol, ncol_all)      # the most recent `ncol` differences (column ORDER does not matter to the candidate)
            dX = np.column_stack([hist[i + 1][0] - hist[i][0] for i in cols])
            dF = np.column_stack([hist[i + 1][1] - hist[i][1] for i in cols])
            dG = np.column_stack([hist[i + 1][2] - hist[i][2] for i in cols])
            if type1:
                M = dX.T @ dF
                eta = np.linalg.solve(M, dX.T @ f)
                cond = np.linalg.cond(M)
            else:
                eta = np.linalg.lstsq(dF, f, rcond=None)[0]
                cond = np.linalg.cond(dF) ** 2            # the normal equations square the condition number
            want = g - dG @ eta
            if cond < 1e10:
                assert np.linalg.norm(cand - want) <= 1e-10 * cond * max(1.0, np.linalg.norm(want)), (it, ncol, cond)
                checked += 1
        x = cand
    return checked, aa


@pytest.mark.parametrize("type1", [True, False])
@pytest.mark.parametrize("rolling", [True, False])
@pytest.mark.parametrize("dim,mem", [(40, 6), (200, 10)])
def test_normal_equation_variants_produce_the_defined_candidate(dim, mem, type1, rolling):
    checked, aa = replay_ne(dim, 3 * mem + 7, mem, seed=dim + mem + 2 * type1 + rolling, type1=type1, rolling=rolling)
    assert checked >= mem
    if rolling:
        assert aa.num_restarts == 0 and aa.iter > 2 * aa.mem          # the rolling memory never empties its history
    else:
        assert aa.num_restarts >= 2


def test_variants_on_a_linear_map_and_through_the_loop():
    """On a linear contraction with memory >= dim every variant reaches the fixed point within dim + 2 steps (both types span the same Krylov space);
    through the ADMM loop the reference's simple QP solves to its golden (simple.jl:45-47) in fewer iterations than the plain loop."""
    dim = 8
    rng = np.random.default_rng(3)
    B = rng.standard_normal((dim, dim)); B = 0.5 * B / np.linalg.norm(B, 2)
    c = rng.standard_normal(dim)
    xs = np.linalg.solve(np.eye(dim) - B, c)
    for t1 in (True, False):
        for roll in (True, False):
            aa = O.AndersonAcceleratorNE(dim, mem=dim, min_mem=1, type1=t1, rolling=roll)
            x = np.zeros(dim); errs = []
            for _ in range(dim + 2):
                g = B @ x + c
                aa.update(g, x); aa.accelerate(g)
                x = g
                errs.append(np.linalg.norm(x - xs))
            assert min(errs) <= 1e-9 * np.linalg.norm(xs)
    P = np.array([[4.0, 1], [1, 2]]); q = np.array([1.0, 1])
    A = np.array([[1.0, 1], [1, 0], [0, 1]]); l = np.array([1.0, 0, 0]); u = np.array([1.0, 0.7, 0.7])
    Am, b, cones = O.assemble([O.Constraint(np.vstack([-A, A]), np.concatenate([u, -l]), O.Nonnegatives(6))])
    plain = O.Workspace(P, q, Am, b, cones, O.Settings()).optimize()
    for name in O.ACCELERATOR_VARIANTS:
        r = O.Workspace(P, q, Am, b, cones, O.Settings(accelerator=name)).optimize()
        assert r.status == "Solved" and abs(r.obj_val - 1.88) < 1e-3 and np.linalg.norm(r.x - [0.3, 0.7]) < 1e-3 and r.iter < plain.iter, name
