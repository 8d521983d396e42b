"""GPU parity tests of the batch mode (BASELINE config 3: independent SOCPs, one persistent workgroup per problem)."""
import numpy as np
import pytest

import cosmo_jl_amd as cj
from oracle import cosmo_oracle as O
from tests import util

pytestmark = pytest.mark.gpu
F = cj._ffi


def _models(probs, st):
    out = []
    for p in probs:
        md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], st)
        out.append(md)
    return out


def test_batch_of_socps_matches_oracle_per_problem():
    probs = [cj.problems.socp(n=60, m=120, ncones=12, nnz=900, seed=100 + k) for k in range(9)]
    st = cj.Settings()
    res = cj.optimize_batch(_models(probs, st))
    for p, r in zip(probs, res):
        ref = O.solve(p["P"], p["q"], p["A"], p["b"], util.oracle_cones(p["sets"]), O.Settings(kkt_solver="cg"))
        assert r.status == ref.status == "Solved"
        assert abs(r.iter - ref.iter) <= 25                        # same count within one check_termination interval
        assert abs(r.obj_val - ref.obj_val) <= 1e-4 * (1 + abs(ref.obj_val))
        assert np.linalg.norm(r.x - ref.x) <= 1e-3 * max(1.0, np.linalg.norm(ref.x))
        assert len(r.info.rho_updates) == len(ref.rho_updates)


def test_batch_tight_mode_trajectory_and_independence():
    # problems with very different scales in one batch: every problem keeps its own rho / CG / status
    rng = np.random.default_rng(0)
    probs = []
    for k in range(5):
        p = util.random_qp(rng, 40, 3, 30, 30, soc_dims=(4, 7), p_shift=5.0)
        scale = 10.0 ** (k - 2)
        p["q"] = p["q"] * scale
        probs.append(p)
    st_o = O.Settings(scaling=10, kkt_solver="cg", tol_constant=1e-10, tol_exponent=0.0, max_iter=120, eps_abs=0.0, eps_rel=0.0,
                      check_infeasibility=10 ** 9)
    st = cj.Settings(kkt_solver=cj.with_options(cj.CGIndirectKKTSolver, tol_constant=1e-10, tol_exponent=0.0), max_iter=120,
                     eps_abs=0.0, eps_rel=0.0)
    res = cj.optimize_batch(_models(probs, st))
    for p, r in zip(probs, res):
        ref = O.solve(p["P"], p["q"], p["A"], p["b"], util.oracle_cones(p["sets"]), st_o)
        assert r.status == ref.status == "Max_iter_reached" and r.iter == ref.iter == 120
        assert np.max(np.abs(r.x - ref.x)) <= 1e-7 * max(np.max(np.abs(ref.x)), 1e-300)
        assert np.max(np.abs(r.s - ref.s)) <= 1e-7 * max(np.max(np.abs(ref.s)), 1.0)
        assert np.max(np.abs(r.y - ref.y)) <= 1e-7 * max(np.max(np.abs(ref.y)), 1.0)
        assert len(r.info.rho_updates) == len(ref.rho_updates)
        assert np.allclose(r.info.rho_updates, ref.rho_updates, rtol=1e-6)
    # a batch of one equals the single-problem path within rounding (different reduction grouping only)
    single = cj.optimize(_models(probs[:1], st)[0])
    assert np.max(np.abs(single.x - res[0].x)) <= 1e-9 * max(np.max(np.abs(res[0].x)), 1e-300)


def test_batch_box_and_zero_cones_and_rho_classes():
    rng = np.random.default_rng(5)
    probs = [util.random_qp(rng, 30, 4, 20, 40) for _ in range(4)]
    # same structure, different data/bounds per problem
    st = cj.Settings()
    mods = _models(probs, st)
    res = cj.optimize_batch(mods)
    for p, r in zip(probs, res):
        ref = O.solve(p["P"], p["q"], p["A"], p["b"], util.oracle_cones(p["sets"]), O.Settings(kkt_solver="cg"))
        assert r.status == ref.status
        assert abs(r.iter - ref.iter) <= 25
        assert abs(r.obj_val - ref.obj_val) <= 1e-4 * (1 + abs(ref.obj_val))
    with pytest.raises(ValueError):
        cj.optimize_batch(_models([probs[0], cj.problems.socp(n=60, m=120, ncones=12, nnz=900)], st))


def test_cfg3_full_size_batch():
    # BASELINE config 3 on one GPU: 1024 independent SOCPs n=500, m=1000, 50 SecondOrderCone(20) each
    probs = [cj.problems.socp(seed=1000 + k) for k in range(1024)]
    st = cj.Settings()
    mods = _models(probs, st)
    res = cj.optimize_batch(mods)
    assert all(r.status == "Solved" for r in res)
    # spot-check a few problems against the oracle
    for k in (0, 511, 1023):
        p = probs[k]
        ref = O.solve(p["P"], p["q"], p["A"], p["b"], util.oracle_cones(p["sets"]), O.Settings(kkt_solver="cg"))
        assert ref.status == "Solved" and abs(res[k].iter - ref.iter) <= 25
        assert abs(res[k].obj_val - ref.obj_val) <= 1e-4 * (1 + abs(ref.obj_val))
    # solutions are feasible: s in the SOCs, A x + s = b to the solver tolerance
    for k in (3, 700):
        r, p = res[k], probs[k]
        s = r.s.reshape(50, 20)
        assert np.all(np.linalg.norm(s[:, 1:], axis=1) <= s[:, 0] + 1e-9)
        assert np.max(np.abs(p["A"] @ r.x + r.s - p["b"])) <= 1e-3
    print("cfg3 batch: iter_time %.3f s for %d problems, iterations min/median/max = %d/%d/%d" % (
        res[0].times.iter_time, len(res), min(r.iter for r in res), int(np.median([r.iter for r in res])), max(r.iter for r in res)))


def _with_env(env, fn):
    import os
    keys = ("COSMO_HIP_BATCH_LDS", "COSMO_HIP_BATCH_BS", "COSMO_HIP_BATCH_REG")
    saved = {k: os.environ.pop(k, None) for k in keys}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k in keys:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]


def test_batch_kernel_variants_agree():
    """The three batch kernels (streaming / LDS image / register-resident iterates) on the same batch: the LDS-image kernel at
    256 threads repeats the streaming kernel bit for bit (same tile order, same row sums, same reduction tree); the
    register-resident kernel only differs in the block-reduction tree (tight-CG trajectories agree to 1e-9)."""
    probs = [cj.problems.socp(seed=2000 + k) for k in range(6)]           # config-3 sized: n = 500, m = 1000, 50 cones
    tight = cj.with_options(cj.CGIndirectKKTSolver, tol_constant=1e-10, tol_exponent=0.0)
    st = cj.Settings(kkt_solver=tight, max_iter=60, eps_abs=0.0, eps_rel=0.0)
    run = lambda: cj.optimize_batch(_models(probs, st))
    ref = _with_env({"COSMO_HIP_BATCH_LDS": "0"}, run)
    lds = _with_env({"COSMO_HIP_BATCH_REG": "0", "COSMO_HIP_BATCH_BS": "256"}, run)
    reg = _with_env({}, run)
    for a, b, c in zip(ref, lds, reg):
        assert np.array_equal(a.x, b.x) and np.array_equal(a.s, b.s) and np.array_equal(a.y, b.y)
        assert a.kkt_iters_total == b.kkt_iters_total and a.info.rho_updates == b.info.rho_updates
        sc = max(1.0, float(np.max(np.abs(a.x))))
        assert np.max(np.abs(a.x - c.x)) <= 1e-9 * sc
        assert np.max(np.abs(a.s - c.s)) <= 1e-9 * max(1.0, float(np.max(np.abs(a.s))))
        assert np.max(np.abs(a.y - c.y)) <= 1e-8 * max(1.0, float(np.max(np.abs(a.y))))
        assert len(a.info.rho_updates) == len(c.info.rho_updates)
        assert a.iter == c.iter == 60 and a.status == c.status


def test_batch_register_kernel_default_schedule_matches_oracle():
    # default (inexact) CG schedule on config-3 sized problems through the register-resident kernel, against the oracle
    probs = [cj.problems.socp(seed=3000 + k) for k in range(3)]
    res = cj.optimize_batch(_models(probs, cj.Settings()))
    for p, r in zip(probs, res):
        ref = O.solve(p["P"], p["q"], p["A"], p["b"], util.oracle_cones(p["sets"]), O.Settings(kkt_solver="cg"))
        assert r.status == ref.status == "Solved"
        assert abs(r.iter - ref.iter) <= 25
        assert abs(r.obj_val - ref.obj_val) <= 1e-4 * (1 + abs(ref.obj_val))
