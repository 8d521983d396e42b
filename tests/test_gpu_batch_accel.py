"""GPU tests of the ACCELERATED loop in batch mode (csrc/batch.hip, batch_admm_body<..., AA = true>): the reference's default
AndersonAccelerator (src/settings.jl:136-138, src/accelerator_interface.jl:58-130; the algorithm itself lives in COSMOAccelerators.jl -- parity
unpinned) for every problem of a batch, all decisions taken by the problem's persistent workgroup.

An accelerated trajectory is not reproducible across summation orders (the least-squares step amplifies rounding), so the batch is compared with the
CPU oracle's accelerated loop and with the single-problem device path the way tests/test_gpu_anderson.py compares those two: same status, same
solution / objective at the solver tolerance, iteration counts within one check interval on a tight KKT solve."""
import numpy as np
import pytest
import scipy.sparse as sp

import cosmo_jl_amd as cj
from oracle import cosmo_oracle as O
from tests import util
from tests import infeasible_instances as INF

pytestmark = pytest.mark.gpu
F = cj._ffi
TIGHT = dict(tol_constant=1e-10, tol_exponent=0.0)


def _models(probs, st, dtype=np.float64):
    out = []
    for p in probs:
        md = cj.Model(dtype=dtype) if dtype is not np.float64 else cj.Model()
        md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], st)
        out.append(md)
    return out


def _oracle(p, **st):
    ws = O.Workspace(p["P"], p["q"], p["A"], p["b"], util.oracle_cones(p["sets"]), O.Settings(kkt_solver="cg", accelerator="anderson", **st))
    return ws, ws.optimize()


def _random_qps(nprob, seed):
    rng = np.random.default_rng(seed)
    return [util.random_qp(rng, 40, 4, 30, 25, soc_dims=(5, 3), p_shift=2.0) for _ in range(nprob)]


@pytest.mark.parametrize("env", [dict(), dict(COSMO_HIP_BATCH_LDS="0")])
def test_accelerated_batch_matches_oracle_and_single_problem_path(env, monkeypatch):
    """48 random QPs with Zero / Nonnegatives / Box / SecondOrderCone rows, tight CG, eps = 1e-7, accelerator on: per problem the status, objective and
    solution of the oracle's accelerated loop and of the single-problem device path; iteration counts within one check interval; the accelerated
    batch needs fewer iterations than the plain batch.  Both kernel variants (LDS image / streaming)."""
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    probs = _random_qps(48, 5)
    st = dict(eps_abs=1e-7, eps_rel=1e-7)
    tight = cj.with_options(cj.CGIndirectKKTSolver, **TIGHT)
    acc = cj.Settings(accelerator=cj.AndersonAccelerator, kkt_solver=tight, **st)
    res = cj.optimize_batch(_models(probs, acc))
    plain = cj.optimize_batch(_models(probs, cj.Settings(kkt_solver=tight, **st)))
    fewer = close = 0
    for k, (p, r) in enumerate(zip(probs, res)):
        ws, ref = _oracle(p, **TIGHT, **st)
        assert r.status == ref.status == "Solved", (k, r.status, ref.status)
        # instances that need thousands of accelerated iterations are chaotic in the iteration count (the oracle itself moves by hundreds under a
        # re-ordered dot product); the short runs agree within one check interval
        if ref.iter <= 500:
            assert 0.5 * ref.iter - 25 <= r.iter <= 2.0 * ref.iter + 25, (k, r.iter, ref.iter)
        close += abs(r.iter - ref.iter) <= 25
        if ref.iter <= 150:
            assert abs(r.iter - ref.iter) <= 25, (k, r.iter, ref.iter)
        assert abs(r.obj_val - ref.obj_val) <= 1e-5 * (1 + abs(ref.obj_val)), k
        assert np.linalg.norm(r.x - ref.x) <= 1e-4 * max(1.0, np.linalg.norm(ref.x)), k
        # (declined steps of an accelerated run: the count moves with the summation order of the inner products -- parity unpinned; 36 against 51 on
        #  one problem after the register kernel's ownership change of round 5, with status / iteration / objective / solution inside their bounds above)
        assert abs(r.safeguarding_iter - ws.safeguarding_iter) <= 5 + 0.4 * ws.safeguarding_iter, (k, r.safeguarding_iter, ws.safeguarding_iter)
        if plain[k].status == "Solved":                        # (the plain loop runs into max_iter = 5000 on the hardest instance)
            assert abs(plain[k].obj_val - r.obj_val) <= 1e-5 * (1 + abs(r.obj_val))
        fewer += r.iter < plain[k].iter
        if k < 6:                                              # the single-problem device path (csrc/anderson.hip) on the same problem
            md = _models([p], acc)[0]
            one = cj.optimize(md)
            assert one.status == "Solved" and abs(one.iter - r.iter) <= 25 + 0.2 * r.iter and abs(one.obj_val - r.obj_val) <= 1e-5 * (1 + abs(r.obj_val))
    assert fewer >= 0.8 * len(probs) and close >= 0.85 * len(probs), (fewer, close)


def test_accelerated_batch_counters_and_api_limits():
    """accel_stats per problem (accelerated > 0, accepted + declined == accelerated with safeguarding, restarts when the memory fills), Result.iter
    counts the safeguarding steps (src/solver.jl:196), mem > 16 is refused, the accelerator must be installed before set_params."""
    probs = _random_qps(6, 9)
    tight = cj.with_options(cj.CGIndirectKKTSolver, **TIGHT)
    st = cj.Settings(accelerator=cj.with_options(cj.AndersonAccelerator, mem=5), kkt_solver=tight, eps_abs=1e-9, eps_rel=1e-9, max_iter=400)
    mods = _models(probs, st)
    B, _ = cj.model.prepare_batch(mods, 0)
    rs = B.optimize()
    a = B.accel_stats()
    its, solves, _ = B.counters()
    for k, r in enumerate(rs):
        assert a["accelerated"][k] > 0 and a["accepted"][k] + a["declined"][k] == a["accelerated"][k]
        assert a["restarts"][k] >= 1                                        # mem = 5: RestartedMemory starts over several times
        assert a["safeguarding_iter"][k] == a["declined"][k] == r.safeguarding_iter
        assert r.iter == its[k] + a["safeguarding_iter"][k]
        assert solves[k] == r.iter + 1                                      # one KKT solve per ADMM step incl. the safeguarding steps + the init step
    B.close()
    p = probs[0]
    B = F.Batch(1, p["P"].shape[0], p["A"].shape[0], 0)
    with pytest.raises(F.CosmoHipError) as e:
        B.set_accelerator(F.ACCEL_ANDERSON, mem=17)
    assert e.value.code == 6                                                # COSMO_HIP_ERR_UNSUPPORTED
    B.close()
    B, _ = cj.model.prepare_batch(_models(probs[:1], cj.Settings()), 0)
    with pytest.raises(F.CosmoHipError) as e:
        B.set_accelerator(F.ACCEL_ANDERSON)
    assert e.value.code == 1                                                # COSMO_HIP_ERR_INVALID
    assert B.accel_stats()["accelerated"][0] == 0                            # no accelerator: zeros
    B.close()


def test_accelerated_batch_rho_adaption_and_activation():
    """Deferred rho updates (update_suggested, src/solver.jl:284-292) restart the accelerator: the rho-update counts of the reference's golden
    (AccelerationTests/max_rho_adaption.jl:19-32) per problem; IterActivation / AccuracyActivation reach the solution; max_iter counts the
    safeguarding steps (src/solver.jl:140,173)."""
    Ps = sp.csc_matrix(np.array([[4.0, 1], [1, 2]])); q = np.array([1.0, 1])
    A = np.array([[1.0, 1], [1, 0], [0, 1]]); l = np.array([1.0, 0, 0]); u = np.array([1.0, 0.7, 0.7])

    def cons(M):
        return [M.Constraint(-A, u, M.Nonnegatives), M.Constraint(A, -l, M.Nonnegatives)] if M is cj else \
               [M.Constraint(-A, u, M.Nonnegatives(3)), M.Constraint(A, -l, M.Nonnegatives(3))]

    def batch(n, **kw):
        mods = []
        for _ in range(n):
            md = cj.Model()
            cj.assemble(md, Ps, q, cons(cj), settings=cj.Settings(**kw))
            mods.append(md)
        return cj.optimize_batch(mods)

    kw = dict(adaptive_rho_interval=25, adaptive_rho_max_adaptions=2, rho=1e-6, eps_abs=1e-6, eps_rel=1e-4)
    rs = batch(3, accelerator=cj.AndersonAccelerator, **kw)
    Ao, bo, cones = O.assemble(cons(O))
    ref = O.Workspace(Ps, q, Ao, bo, cones, O.Settings(accelerator="anderson", kkt_solver="cg", **kw)).optimize()
    for r in rs:
        assert r.status == ref.status == "Solved" and len(r.info.rho_updates) - 1 == 2 == len(ref.rho_updates) - 1
        assert np.allclose(r.info.rho_updates, ref.rho_updates, rtol=1e-5)
        assert abs(r.obj_val - 1.88) < 1e-3 and np.linalg.norm(r.x - [0.3, 0.7]) < 1e-3              # simple.jl:45-47
    for act in (10, cj.AccuracyActivation(1e-2)):
        r = batch(2, accelerator=cj.AndersonAccelerator, accelerator_activation=act)[1]
        assert r.status == "Solved" and abs(r.obj_val - 1.88) < 1e-3
    tight = dict(kkt_solver=cj.with_options(cj.CGIndirectKKTSolver, **TIGHT))
    r = batch(2, accelerator=cj.AndersonAccelerator, max_iter=20, eps_abs=1e-12, eps_rel=1e-12, **tight)[0]
    ws = O.Workspace(Ps, q, Ao, bo, cones, O.Settings(accelerator="anderson", kkt_solver="cg", max_iter=20, eps_abs=1e-12, eps_rel=1e-12, **TIGHT))
    ref = ws.optimize()
    assert r.iter in (20, 21) and r.status == ("Max_iter_reached" if r.iter == 20 else "Undetermined") or (r.iter, r.status) == (ref.iter, ref.status)
    assert abs(r.iter - ref.iter) <= 1


@pytest.mark.parametrize("family,seed", [(f, s_) for (f, s_) in INF.CASES if f in ("primal_infeasible_1", "dual_infeasible_1")])
def test_accelerated_batch_infeasible_families(family, seed):
    """The reference's infeasible-by-construction LP families in an ACCELERATED batch next to a feasible problem of the same shape: the certificates of
    a problem run at ITS first non-accelerated iteration after a flagged one (the workgroup leaves its launch for them), the feasible neighbour is
    solved regardless.  Status as the oracle's accelerated loop."""
    gen, accepted, _ = INF.FAMILIES[family]
    P, q, cons = gen(seed)
    st = dict(max_iter=2000, eps_abs=1e-5, eps_rel=1e-5)
    kinds = {INF.ZERO: cj.ZeroSet, INF.NONNEG: cj.Nonnegatives, INF.SOC: cj.SecondOrderCone}
    settings = cj.Settings(accelerator=cj.AndersonAccelerator, kkt_solver=cj.with_options(cj.CGIndirectKKTSolver, **TIGHT), **st)
    md = cj.Model()
    cj.assemble(md, P, q, [cj.Constraint(A, b, kinds[k]) for (A, b, k, d) in cons], settings=settings)
    # a feasible problem with the same dimensions and cone structure: the same rows with right-hand sides that admit x = 0, objective 1/2 |x|^2 - 1'x
    n = P.shape[0]
    feas = cj.Model()
    fc, fo_c = [], []
    for (A, b, k, d) in cons:
        bb = np.zeros(d)
        if k == INF.NONNEG:
            bb[:] = 1.0
        elif k == INF.SOC:
            bb[0] = 1.0
        fc.append(cj.Constraint(A, bb, kinds[k]))
        fo_c.append(O.Constraint(A, bb, O.Cone(k, d, constr_type=(np.zeros(d, dtype=bool) if k == O.NONNEG else None))))
    cj.assemble(feas, sp.identity(n, format="csc"), -np.ones(n), fc, settings=settings)
    res = cj.optimize_batch([md, feas])
    # Reference point: the single-problem device path (csrc/anderson.hip + optimize_accelerated).  On these diverging iterates the accelerated loop is
    # chaotic -- the CPU oracle's accelerated loop detects seed 3 of the dual family at iteration 253 and runs into max_iter on seeds 1 and 2; the
    # single-problem device path and the LDS-image batch kernel detect all three, the register batch kernel two of them -- so on the dual family
    # every path only has to be right WHERE it decides; the primal family is decided by all of them at (nearly) the same iteration.
    one = cj.Model()
    cj.assemble(one, P, q, [cj.Constraint(A, b, kinds[k]) for (A, b, k, d) in cons], settings=settings)
    r1 = cj.optimize(one)
    undecided = ("Max_iter_reached", "Undetermined")
    if family == "primal_infeasible_1":
        assert res[0].status == r1.status and res[0].status in accepted, (res[0].status, r1.status)
    else:                                                  # diverging x: whether an accelerated run decides within max_iter depends on rounding (see above)
        assert res[0].status in accepted + undecided and r1.status in accepted + undecided, (res[0].status, r1.status)
    Ao, bo, cones = O.assemble([O.Constraint(A, b, O.Cone(k, d, constr_type=(np.zeros(d, dtype=bool) if k == O.NONNEG else None))) for (A, b, k, d) in cons])
    ref = O.solve(P, q, Ao, bo, cones, O.Settings(kkt_solver="cg", accelerator="anderson", **TIGHT, **st))
    assert ref.status in accepted + undecided
    if family == "primal_infeasible_1":
        assert ref.status == res[0].status and abs(res[0].iter - r1.iter) <= 40 and abs(res[0].iter - ref.iter) <= 80, (res[0].iter, r1.iter, ref.iter)
    fo = O.solve(sp.identity(n, format="csc"), -np.ones(n), *O.assemble(fo_c), O.Settings(kkt_solver="cg", accelerator="anderson", **TIGHT, **st))
    assert res[1].status == fo.status == "Solved" and abs(res[1].obj_val - fo.obj_val) <= 1e-4 * (1 + abs(fo.obj_val))


def test_accelerated_batch_of_small_sdps():
    """PSD cones (side <= 16) inside the accelerated batch loop: same status / objective as the oracle's accelerated loop."""
    rng = np.random.default_rng(77)
    probs = [util.random_qp(rng, 30, 2, 8, 6, soc_dims=(4,), psd_tri_dims=(5, 9), p_shift=1.0) for _ in range(6)]
    st = dict(eps_abs=1e-6, eps_rel=1e-6)
    res = cj.optimize_batch(_models(probs, cj.Settings(accelerator=cj.AndersonAccelerator, kkt_solver=cj.with_options(cj.CGIndirectKKTSolver, **TIGHT), **st)))
    for k, (p, r) in enumerate(zip(probs, res)):
        _, ref = _oracle(p, **TIGHT, **st)
        assert r.status == ref.status, (k, r.status, ref.status)
        if ref.status == "Solved":
            assert abs(r.obj_val - ref.obj_val) <= 1e-4 * (1 + abs(ref.obj_val)), k
            assert abs(r.iter - ref.iter) <= 50, (k, r.iter, ref.iter)


def test_accelerated_batch_float32():
    """libcosmo_hip_f32.so: the accelerated batch loop instantiated for float."""
    probs = _random_qps(8, 31)
    st = dict(eps_abs=1e-4, eps_rel=1e-4)
    res = cj.optimize_batch(_models(probs, cj.Settings(accelerator=cj.AndersonAccelerator, **st), dtype=np.float32))
    for k, (p, r) in enumerate(zip(probs, res)):
        ref = O.solve(p["P"], p["q"], p["A"], p["b"], util.oracle_cones(p["sets"]), O.Settings(kkt_solver="cg", **st))
        assert r.status == "Solved", (k, r.status)
        assert abs(r.obj_val - ref.obj_val) <= 1e-2 * (1 + abs(ref.obj_val)), k
