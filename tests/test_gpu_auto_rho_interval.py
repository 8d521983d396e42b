"""settings.adaptive_rho_interval == 0: the reference's AUTOMATIC rho interval (/root/reference/src/solver.jl:244-256) -- once the loop has run for
adaptive_rho_fraction * setup_time seconds the interval is fixed ONCE to round_multiple(iter, check_termination) (>= check_termination) and written
into the settings; from then on the schedule is that of a fixed interval.  The rule reads the wall clock, so its trigger point is not reproducible
in general; the two ends ARE: setup_time = 0 fires at the first test (interval = check_termination), a huge setup_time never fires.  At both ends the
device run must equal -- bit for bit, same kernels in the same order -- the run with the corresponding fixed setting, and the oracle's literal
restatement of the rule must give the interval the device reports (rejected with COSMO_HIP_ERR_UNSUPPORTED until round 5)."""
import numpy as np
import pytest

import cosmo_jl_amd as cj
from oracle import cosmo_oracle as O
from tests import util

F = cj._ffi


def _prob():
    return util.random_qp(np.random.default_rng(21), 60, 5, 40, 50, soc_dims=(4, 7), p_shift=2.0)


def test_oracle_restatement_of_the_rule():
    assert [O.round_multiple(x, 25) for x in (1, 12, 13, 37, 38, 50, 63)] == [0, 0, 25, 25, 50, 50, 75]       # algebra.jl:245-247
    prob = _prob()

    def run(**kw):
        ws = O.Workspace(prob["P"], prob["q"], prob["A"], prob["b"], util.oracle_cones(prob["sets"]),
                         O.Settings(kkt_solver="cg", tol_constant=1e-10, tol_exponent=0.0, max_iter=400, eps_abs=1e-9, eps_rel=1e-9, **kw))
        return ws
    a = run(adaptive_rho_interval=0); a.setup_time = 0.0
    ra = a.optimize()
    assert a.st.adaptive_rho_interval == 25                         # fired at iteration 1: max(round_multiple(1, 25), 25)
    rb = run(adaptive_rho_interval=25).optimize()
    assert ra.iter == rb.iter and np.array_equal(ra.x, rb.x) and ra.rho_updates == rb.rho_updates and len(ra.rho_updates) >= 2
    c = run(adaptive_rho_interval=0); c.setup_time = 1e9
    rc = c.optimize()
    rd = run(adaptive_rho=False).optimize()
    assert c.st.adaptive_rho_interval == 0 and len(rc.rho_updates) == 1 and np.array_equal(rc.x, rd.x)
    # an injected clock: the condition becomes true during iteration 60 => interval round_multiple(60, 25) = 50
    e = run(adaptive_rho_interval=0); e.setup_time = 1.0
    ticks = iter(range(10 ** 6))
    e.clock = lambda: next(ticks) * (0.4 / 59.5)                    # the k-th reading is taken at the top of iteration k (reading 0 = iter_start)
    e.optimize()
    assert e.st.adaptive_rho_interval == 50


@pytest.mark.gpu
@pytest.mark.parametrize("accel", [False, True])
def test_device_automatic_interval_equals_the_fixed_interval_runs(accel):
    prob = _prob()
    kkt = cj.with_options(cj.CGIndirectKKTSolver, tol_constant=1e-10, tol_exponent=0.0)

    def solve(setup_time=None, **kw):
        st = cj.Settings(kkt_solver=kkt, max_iter=400, eps_abs=1e-9, eps_rel=1e-9, accelerator=(cj.AndersonAccelerator if accel else cj.EmptyAccelerator), **kw)
        md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], st)
        if setup_time is None:
            return cj.optimize(md), md
        cj.model.setup(md)                                           # the steps of optimize(), with a chosen setup_time
        h = md.handle
        h.set_iterates(md.x, md.s, md.mu)
        h.set_setup_time(setup_time)
        r = h.optimize()
        return (r, h.get_iterates()[0], h.rho_interval()), md

    (r0, w0, ri0), _ = solve(setup_time=0.0, adaptive_rho_interval=0)
    assert ri0[0] == 25 and ri0[1] in (1, 2)                        # fixed at the first test of the loop
    (r1, w1, _), _ = solve(setup_time=0.0, adaptive_rho_interval=25)
    assert r0.iter == r1.iter and r0.status == r1.status and r0.n_rho_updates == r1.n_rho_updates >= (1 if accel else 2)   # (the accelerated run is Solved before rho moves)
    assert np.array_equal(w0, w1)                                   # the same launches in the same order
    (r2, w2, ri2), _ = solve(setup_time=1e9, adaptive_rho_interval=0)
    (r3, w3, _), _ = solve(setup_time=0.0, adaptive_rho=False)
    assert ri2 == (0, -1) and r2.n_rho_updates == 1 and r2.iter == r3.iter and np.array_equal(w2, w3)
    # through the mirrored front-end: setup_time is measured, the chosen interval is written back into the settings (solver.jl:249-254)
    res, md = solve(adaptive_rho_interval=0, adaptive_rho_fraction=0.0)
    assert md.settings.adaptive_rho_interval == 25 and res.status in ("Solved", "Max_iter_reached")


@pytest.mark.gpu
def test_batch_kernels_refuse_the_automatic_interval_and_optimize_batch_runs_it_on_handles():
    """cosmo_hip_batch_* has no host clock inside its persistent kernels and refuses adaptive_rho_interval = 0; optimize_batch sends such a list through
    the batch group, where every problem gets its own handle and the single-problem rule (csrc/batch_group.hip)."""
    prob = _prob()
    st = cj.Settings(adaptive_rho_interval=0)

    def models():
        ms = []
        for _ in range(2):
            md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], st); ms.append(md)
        return ms
    with pytest.raises(Exception, match="adaptive_rho_interval"):
        cj.model.prepare_batch(models(), 0)
    res = cj.optimize_batch(models())
    md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], cj.Settings(adaptive_rho_interval=0))
    one = cj.optimize(md)
    for r in res:
        assert r.status == one.status == "Solved" and abs(r.obj_val - one.obj_val) <= 1e-4 * (1 + abs(one.obj_val))
