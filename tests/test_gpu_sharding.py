"""GPU tests of the clique-sharded projection on ONE device: ownership ranges + slice merge reproduce the full
projection bit for bit; the RCCL path is exercised with a single-rank communicator; the exchange step itself
(comm_enqueue_exchange with nranks = 2: row_lo / row_hi slices, exchange point of the iteration) runs in TWO PROCESSES that share
the device through the host-staged transport and must reproduce the single-rank run bit for bit."""
import os
import subprocess
import sys
import tempfile
import uuid

import numpy as np
import pytest
import scipy.sparse as sp

import cosmo_jl_amd as cj
from oracle import cosmo_oracle as O
from tests import util

pytestmark = pytest.mark.gpu
F = cj._ffi


def _handle(sets):
    m = sum(K.dim for K in sets)
    h = cj.Handle(0)
    h.set_problem(sp.identity(2, format="csc"), np.zeros(2), sp.csc_matrix((m, 2)), np.zeros(m))
    h.set_cones([K.kind for K in sets], [K.dim for K in sets], None, None)
    return h, m


def test_sharded_projection_merges_to_the_full_projection():
    rng = np.random.default_rng(11)
    dims = [5, 40, 17, 90, 8, 33, 130, 12]
    sets = [cj.Nonnegatives(7), cj.SecondOrderCone(6)] + [cj.PsdConeTriangle(d * (d + 1) // 2) for d in dims] + [cj.SecondOrderCone(9)]
    h, m = _handle(sets)
    s = rng.standard_normal(m)
    full, ranks_full, br_full = h.project(s)
    world = 3
    bounds = cj.partition_cones_contiguous(cj.cone_costs(sets), world)
    offs = np.concatenate([[0], np.cumsum([K.dim for K in sets])])
    merged = np.full(m, np.nan)
    for r in range(world):
        hr, _ = _handle(sets)
        hr.set_cone_ownership(bounds[r], bounds[r + 1])
        part, ranks, br = hr.project(s)
        lo, hi = offs[bounds[r]], offs[bounds[r + 1]]
        merged[lo:hi] = part[lo:hi]                       # what the owner broadcasts
        for k in range(len(sets)):                         # ranks / branches are reported only for owned cones
            if bounds[r] <= k < bounds[r + 1]:
                assert ranks[k] == ranks_full[k] and br[k] == br_full[k]
            elif sets[k].kind in (F.PSD_TRIANGLE, F.SOC):
                assert ranks[k] == -1 and br[k] == -1
        # rows of cones owned by other ranks are left untouched (Nonnegatives rows are projected by everyone)
        other = np.ones(m, dtype=bool); other[lo:hi] = False; other[:7] = False
        assert np.array_equal(part[other], s[other])
    assert np.array_equal(merged.view(np.int64), full.view(np.int64))   # bit for bit


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_single_rank_communicator_rccl_path(dtype):
    """Both libraries: the broadcast's element type follows `cosmo_hip_real` (ncclFloat64 / ncclFloat32)."""
    prob = cj.problems.chordal_sdp(ncliques=8, dmin=4, dmax=30, sep_min=1, sep_max=3, n_total=600, n_zero=5, n_nonneg=10)
    st = cj.Settings(max_iter=100, eps_abs=0, eps_rel=0)
    ref_model = cj.Model(dtype=dtype); ref_model.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], st)
    ref = cj.optimize(ref_model)
    model = cj.Model(dtype=dtype); model.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], st)
    cj.model.setup(model)
    uid = cj.Handle.comm_unique_id()
    assert len(uid) == 128
    model.handle.comm_init(0, 1, uid)
    model.handle.set_cone_shard(cj.partition_cones_contiguous(cj.cone_costs(model.sets), 1))
    model.handle.set_iterates(model.x, model.s, model.mu)
    model.handle.comm_selftest()                                   # ncclBroadcast on the handle's stream
    chk = model.handle.comm_allreduce_check(model.n)               # known-answer all-reduce through the loop's exchange path (1 rank: identity)
    assert chk["exact_mismatches"] == 0 and chk["inexact_outside_bound"] == 0 and chk["transport"] == 1 and chk["nranks"] == 1 and chk["rccl_version_code"] > 20000
    res = cj.optimize(model)
    assert res.status == ref.status and res.iter == ref.iter
    assert np.array_equal(res.x, ref.x) and np.array_equal(res.s, ref.s)


def test_infeasibility_certificates_in_sharded_runs():
    """Clique-sharded runs keep the certificates: every rank tests the cones it owns and the violation flags are max-reduced
    over the communicator (csrc/comm.hip: comm_allreduce_flag).  Exercised here with a single-rank communicator."""
    cases = []
    # primal infeasible SDP (3x3, svec variables): X psd and X11 = -1 ; plus an SOC block so that both cone kinds are owned
    nt = 6
    A = sp.vstack([sp.csc_matrix(([1.0], ([0], [0])), shape=(1, nt + 3)), sp.hstack([sp.identity(nt), sp.csc_matrix((nt, 3))]),
                   sp.hstack([sp.csc_matrix((3, nt)), sp.identity(3)])], format="csc")
    b = np.concatenate([[1.0], np.zeros(nt + 3)])
    cases.append((sp.csc_matrix((nt + 3, nt + 3)), np.zeros(nt + 3), [cj.Constraint(A[:1], b[:1], cj.ZeroSet), cj.Constraint(A[1:1 + nt], b[1:1 + nt], cj.PsdConeTriangle),
                                                                       cj.Constraint(A[1 + nt:], b[1 + nt:], cj.SecondOrderCone)], "Primal_infeasible"))
    # dual infeasible SOCP: minimise -t over the cone
    cases.append((sp.csc_matrix((3, 3)), np.array([-1.0, 0.0, 0.0]), [cj.Constraint(sp.identity(3, format="csc"), np.zeros(3), cj.SecondOrderCone)], "Dual_infeasible"))
    for P, q, cons, want in cases:
        ref_model = cj.Model(); cj.assemble(ref_model, P, q, cons, settings=cj.Settings())
        ref = cj.optimize(ref_model)
        model = cj.Model(); cj.assemble(model, P, q, cons, settings=cj.Settings())
        cj.model.setup(model)
        model.handle.comm_init(0, 1, cj.Handle.comm_unique_id())
        model.handle.set_cone_shard(cj.partition_cones_contiguous(cj.cone_costs(model.sets), 1))
        res = cj.optimize(model)
        assert ref.status == want and res.status == want and res.iter == ref.iter


# ---------------------------------------------------------------------------------------------------------------------
# two ranks = two processes on the one GPU of the box
# ---------------------------------------------------------------------------------------------------------------------
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "shard_worker.py")
ITERS = 60


def _spawn(transport, world, rdv, tmp, timeout=240, extra_env=None):
    env = dict(os.environ, COSMO_HIP_POLAR_KLIFT="10")            # pin the sign-iteration schedule (it is pinned in sharded runs anyway)
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, WORKER, transport, str(r), str(world), rdv, os.path.join(tmp, "rank%d.npz" % r), str(ITERS)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            if transport == "rccl":
                return [(-9, "timed out (RCCL communicator setup with two ranks on one device did not return)")]
            raise
        outs.append((p.returncode, o))
    return outs


def _worker_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("shard_worker", WORKER)
    W = importlib.util.module_from_spec(spec); spec.loader.exec_module(W)
    return W


def _single_rank_reference(monkeypatch, tight=False):
    W = _worker_module()
    monkeypatch.setenv("COSMO_HIP_POLAR_KLIFT", "10")
    p = W.problem()
    md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], W.settings(ITERS, tight))
    return cj.optimize(md), md


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_with_several_ranks_is_bit_identical_to_the_single_rank_run(world, monkeypatch):
    ref, md = _single_rank_reference(monkeypatch)
    bounds = cj.partition_cones_contiguous(cj.cone_costs(md.sets), world)
    assert all(bounds[r + 1] > bounds[r] for r in range(world))                    # every rank owns cliques
    with tempfile.TemporaryDirectory() as tmp:
        outs = _spawn("shm", world, "/cosmo_test_" + uuid.uuid4().hex[:12], tmp)
        for rc, o in outs:
            assert rc == 0, o[-2000:]
        for r in range(world):
            z = np.load(os.path.join(tmp, "rank%d.npz" % r))
            assert int(z["nranks"]) == world and int(z["transport"]) == 2
            assert int(z["exchanges"]) >= ITERS                                    # one exchange step per iteration really ran with nranks > 1
            assert np.array_equal(z["bounds"], np.array(bounds))
            assert int(z["iter"]) == ref.iter == ITERS and int(z["kkt"]) == ref.kkt_iters_total
            for key, val in (("x", ref.x), ("s", ref.s), ("y", ref.y)):
                assert np.array_equal(z[key].view(np.int64), val.view(np.int64)), (r, key)       # bit for bit, on every rank
            assert float(z["obj"]) == ref.obj_val and float(z["r_prim"]) == ref.info.r_prim


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_run_agrees_with_the_single_rank_run(world, monkeypatch):
    """Row sharding (csrc/rowshard.hip, SURVEY 8e option 2 on a replicated CG): every rank keeps its cones AND their rows of A / s / mu /
    rho; the one exchange of an iteration is the all-reduce of the n-vector A'(rho .* ls_s) (kktsolver_indirect.jl:52-54).  The summation
    order of that vector changes with the partition, so the run is compared with the single-rank run at the tight-CG trajectory tolerance
    (1e-7) -- but the ranks must agree with EACH OTHER bit for bit (their control flow depends on it)."""
    ref, md = _single_rank_reference(monkeypatch, tight=True)
    bounds = cj.partition_cones_contiguous(cj.model.row_shard_costs(md.sets), world)
    assert all(bounds[r + 1] > bounds[r] for r in range(world))
    n, m = md.n, md.m
    with tempfile.TemporaryDirectory() as tmp:
        outs = _spawn("shm", world, "/cosmo_test_" + uuid.uuid4().hex[:12], tmp, extra_env={"COSMO_TEST_SHARD": "rows", "COSMO_TEST_TIGHT": "1"})
        for rc, o in outs:
            assert rc == 0, o[-2000:]
        zs = [np.load(os.path.join(tmp, "rank%d.npz" % r)) for r in range(world)]
        rows = 0
        for r, z in enumerate(zs):
            assert int(z["nranks"]) == world and int(z["transport"]) == 2 and str(z["mode"]) == "rows"
            assert np.array_equal(z["bounds"], np.array(bounds))
            rows += int(z["row_hi"]) - int(z["row_lo"])
            # one all-reduce of n reals per KKT solve (init step + ITERS iterations) + one of n + 2 world per residual check / rho rule
            assert int(z["allreduces"]) >= ITERS + 1 and int(z["allreduce_elems"]) in (n, n + 2 * world)
            assert (ITERS + 1) * 8 * n <= int(z["bytes"]) <= (ITERS + 12) * 8 * (n + 2 * world) + 4 * 8 * (n + m)   # + the final all-gathers of get_iterates
            assert int(z["iter"]) == ref.iter == ITERS and str(z["status"]) == ref.status
            assert abs(int(z["kkt"]) - ref.kkt_iters_total) <= max(8, ref.kkt_iters_total // 50)
            for key, val in (("x", ref.x), ("s", ref.s), ("y", ref.y)):
                assert np.max(np.abs(z[key] - val)) <= 1e-7 * max(np.max(np.abs(val)), 1e-30), (r, key)
            assert abs(float(z["obj"]) - ref.obj_val) <= 1e-7 * (1 + abs(ref.obj_val))
            assert len(z["rho_updates"]) == len(ref.info.rho_updates)
        assert rows == m                                                          # the ranks' row ranges tile [0, m)
        for z in zs[1:]:                                                           # identical bits on every rank
            for key in ("x", "s", "y"):
                assert np.array_equal(z[key].view(np.int64), zs[0][key].view(np.int64)), key
            assert float(z["obj"]) == float(zs[0]["obj"]) and float(z["r_prim"]) == float(zs[0]["r_prim"]) and float(z["r_dual"]) == float(zs[0]["r_dual"])
            assert int(z["kkt"]) == int(zs[0]["kkt"])


def test_row_sharded_run_in_float32(monkeypatch):
    """The Float32 library (libcosmo_hip_f32.so): the all-reduce / all-gather element type follows cosmo_hip_real.  Two ranks against the
    single-rank Float32 run at a Float32 tolerance (default CG schedule), the ranks bit-identical to each other."""
    W = _worker_module()
    monkeypatch.setenv("COSMO_HIP_POLAR_KLIFT", "5")
    p = W.problem()
    md = cj.Model(dtype=np.float32); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], W.settings(ITERS))
    ref = cj.optimize(md)
    with tempfile.TemporaryDirectory() as tmp:
        outs = _spawn("shm", 2, "/cosmo_test_" + uuid.uuid4().hex[:12], tmp, extra_env={"COSMO_TEST_SHARD": "rows", "COSMO_TEST_DTYPE": "float32", "COSMO_HIP_POLAR_KLIFT": "5"})
        for rc, o in outs:
            assert rc == 0, o[-2000:]
        zs = [np.load(os.path.join(tmp, "rank%d.npz" % r)) for r in range(2)]
        for z in zs:
            assert z["x"].dtype == np.float32 and str(z["mode"]) == "rows" and int(z["iter"]) == ref.iter == ITERS
            assert int(z["allreduce_elems"]) in (md.n, md.n + 4) and int(z["bytes"]) >= (ITERS + 1) * 4 * md.n      # 4-byte elements on the wire
            for key, val in (("x", ref.x), ("s", ref.s)):
                assert np.max(np.abs(z[key] - val)) <= 2e-3 * max(np.max(np.abs(val)), 1e-30), key
        for key in ("x", "s", "y"):
            assert np.array_equal(zs[0][key].view(np.int32), zs[1][key].view(np.int32)), key


@pytest.mark.parametrize("case,want", [("pinf", "Primal_infeasible"), ("dinf", "Dual_infeasible")])
def test_row_sharded_run_keeps_the_infeasibility_certificates(case, want, monkeypatch):
    """Two ranks, default settings: ||E dy|| / <dy, b> / the support function are reduced over the ranks, A' dy is all-reduced, every rank
    tests its own cones (infeas.hip) -- same status and iteration count as the single-rank run (src/infeasibility.jl:1-68)."""
    W = _worker_module()
    ref = cj.optimize(W.infeasible_model(case))
    assert ref.status == want
    with tempfile.TemporaryDirectory() as tmp:
        outs = _spawn("shm", 2, "/cosmo_test_" + uuid.uuid4().hex[:12], tmp, extra_env={"COSMO_TEST_SHARD": "rows", "COSMO_TEST_CASE": case})
        for rc, o in outs:
            assert rc == 0, o[-2000:]
        for r in range(2):
            z = np.load(os.path.join(tmp, "rank%d.npz" % r))
            assert str(z["mode"]) == "rows" and int(z["row_hi"]) > int(z["row_lo"])
            assert str(z["status"]) == want and int(z["iter"]) == ref.iter


def test_rccl_two_ranks_on_one_device_is_refused_or_identical(monkeypatch):
    """RCCL normally rejects two ranks on one device ("Duplicate GPU detected"); when it does, this test records that as a skip
    (the multi-rank RCCL path needs a multi-GPU node: the driver's scaling run).  Should a build accept it, the result must be
    bit-identical to the single-rank run as well."""
    ref, md = _single_rank_reference(monkeypatch)
    with tempfile.TemporaryDirectory() as tmp:
        outs = _spawn("rccl", 2, os.path.join(tmp, "uid.bin"), tmp, timeout=60)
        if any(rc != 0 for rc, _ in outs):
            msg = " | ".join(o.strip().splitlines()[-1] if o.strip() else "" for _, o in outs)
            pytest.skip("RCCL refused 2 ranks on one device: " + msg[-300:])
        for r in range(2):
            z = np.load(os.path.join(tmp, "rank%d.npz" % r))
            assert int(z["transport"]) == 1 and int(z["exchanges"]) >= ITERS
            for key, val in (("x", ref.x), ("s", ref.s), ("y", ref.y)):
                assert np.array_equal(z[key].view(np.int64), val.view(np.int64)), (r, key)


@pytest.mark.parametrize("mode,world", [("rows", 2), ("rows", 3), ("cones", 2)])
def test_sharded_runs_with_the_reference_default_accelerator(mode, world, monkeypatch):
    """Anderson acceleration (the reference's default, src/settings.jl:136-138; src/accelerator_interface.jl:58-116) in sharded runs, two ranks.
    rows:  w = [x ; the rank's rows]: the accelerator's inner products are all-reduced partial sums (csrc/anderson.hip) -- the ranks agree with each
           other bit for bit (same R, eta, success and safeguarding decisions), and with the single-rank accelerated run within the f2 tolerances
           (same status, iteration count within one check interval, objective 1e-5; acceleration is sensitive to rounding, the summation order of
           A'y and of the inner products changes with the partition);
    cones: every rank holds the whole w, all scalars are redundant and bit-identical => identical to the single-rank run, bit for bit."""
    monkeypatch.setenv("COSMO_TEST_ACCEL", "1")
    monkeypatch.setenv("COSMO_HIP_POLAR_KLIFT", "10")
    W = _worker_module()
    p = W.problem()
    md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], W.settings(3000))
    ref = cj.optimize(md)
    racc = md.handle.accel_stats()
    assert ref.status == "Solved" and racc["accelerated"] > 0
    with tempfile.TemporaryDirectory() as tmp:
        global ITERS
        keep = ITERS
        ITERS = 3000
        try:
            outs = _spawn("shm", world, "/cosmo_test_" + uuid.uuid4().hex[:12], tmp, timeout=400, extra_env={"COSMO_TEST_SHARD": mode, "COSMO_TEST_ACCEL": "1"})
        finally:
            ITERS = keep
        for rc, o in outs:
            assert rc == 0, o[-3000:]
        z = [np.load(os.path.join(tmp, "rank%d.npz" % r)) for r in range(world)]
    for r in range(1, world):
        for key in ("x", "s", "y"):
            assert np.array_equal(z[0][key].view(np.int64), z[r][key].view(np.int64)), (r, key)           # the ranks agree bit for bit
        assert int(z[0]["iter"]) == int(z[r]["iter"]) and int(z[0]["accelerated"]) == int(z[r]["accelerated"]) > 0 and int(z[0]["declined"]) == int(z[r]["declined"])
    assert str(z[0]["status"]) == ref.status
    if mode == "cones":
        assert int(z[0]["iter"]) == ref.iter and int(z[0]["accelerated"]) == racc["accelerated"]
        for key, val in (("x", ref.x), ("s", ref.s), ("y", ref.y)):
            assert np.array_equal(z[0][key].view(np.int64), val.view(np.int64)), key
    else:
        assert abs(int(z[0]["iter"]) - ref.iter) <= 25, (int(z[0]["iter"]), ref.iter)
        assert abs(float(z[0]["obj"]) - ref.obj_val) <= 1e-5 * (1 + abs(ref.obj_val))
        assert np.linalg.norm(z[0]["x"] - ref.x) <= 1e-3 * max(1.0, np.linalg.norm(ref.x))
        assert int(z[0]["allreduces"]) >= int(z[0]["iter"])                                                  # the loop's own all-reduce ran every iteration

def test_row_sharded_run_with_the_type1_rolling_accelerator(monkeypatch):
    """The Type1 / RollingMemory variant (docs/src/acceleration.md:23) on a row-sharded handle: the 2 l inner products per update that keep M = X' F
    current and the l products of X' f are all-reduced partial sums like the default variant's -- two ranks agree with each other bit for bit and
    with the single-rank run within the f2 tolerances."""
    monkeypatch.setenv("COSMO_TEST_ACCEL", "1")
    monkeypatch.setenv("COSMO_TEST_ACCEL_VARIANT", "type1_rolling")
    monkeypatch.setenv("COSMO_HIP_POLAR_KLIFT", "10")
    W = _worker_module()
    p = W.problem()
    md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], W.settings(3000))
    assert md.settings.accelerator.accel_kind == cj._ffi.ACCEL_ANDERSON_TYPE1_ROLLING
    ref = cj.optimize(md)
    racc = md.handle.accel_stats()
    assert ref.status == "Solved" and racc["accelerated"] > 0 and racc["restarts"] == 0
    with tempfile.TemporaryDirectory() as tmp:
        global ITERS
        keep = ITERS
        ITERS = 3000
        try:
            outs = _spawn("shm", 2, "/cosmo_test_" + uuid.uuid4().hex[:12], tmp, timeout=400,
                          extra_env={"COSMO_TEST_SHARD": "rows", "COSMO_TEST_ACCEL": "1", "COSMO_TEST_ACCEL_VARIANT": "type1_rolling"})
        finally:
            ITERS = keep
        for rc, o in outs:
            assert rc == 0, o[-3000:]
        z = [np.load(os.path.join(tmp, "rank%d.npz" % r)) for r in range(2)]
    for key in ("x", "s", "y"):
        assert np.array_equal(z[0][key].view(np.int64), z[1][key].view(np.int64)), key
    assert int(z[0]["iter"]) == int(z[1]["iter"]) and int(z[0]["accelerated"]) == int(z[1]["accelerated"]) > 0 and int(z[0]["declined"]) == int(z[1]["declined"])
    assert str(z[0]["status"]) == ref.status
    assert abs(float(z[0]["obj"]) - ref.obj_val) <= 1e-5 * (1 + abs(ref.obj_val))
    assert np.linalg.norm(z[0]["x"] - ref.x) <= 1e-3 * max(1.0, np.linalg.norm(ref.x))



def test_row_sharded_run_with_a_time_limit_stops_all_ranks_at_the_same_iteration():
    """settings.time_limit in a sharded run (src/solver.jl:351-354): the ranks' clocks differ, so every time-limit decision (the slice a rank enqueues
    and the test itself) is taken on the maximum of the ranks' elapsed seconds -- both ranks return Time_limit_reached at the SAME iteration with the
    same iterates, and the collectives stay matched (no hang)."""
    with tempfile.TemporaryDirectory() as tmp:
        global ITERS
        keep = ITERS
        ITERS = 10 ** 6
        try:
            outs = _spawn("shm", 2, "/cosmo_test_" + uuid.uuid4().hex[:12], tmp, timeout=300, extra_env={"COSMO_TEST_SHARD": "rows", "COSMO_TEST_TIMELIMIT": "1.0"})
        finally:
            ITERS = keep
        for rc, o in outs:
            assert rc == 0, o[-3000:]
        z = [np.load(os.path.join(tmp, "rank%d.npz" % r)) for r in range(2)]
    assert str(z[0]["status"]) == str(z[1]["status"]) == "Time_limit_reached"
    assert int(z[0]["iter"]) == int(z[1]["iter"]) > 0
    for key in ("x", "s", "y"):
        assert np.array_equal(z[0][key].view(np.int64), z[1][key].view(np.int64)), key


def test_row_sharded_auto_rho_interval_fires_on_all_ranks_together_when_their_setup_times_differ():
    """adaptive_rho_interval = 0 (the automatic interval, src/solver.jl:244-256) in a row-sharded run where the ranks were GIVEN different
    ws.times.setup_time (rank 0: 0 s -- its rule would fire at the first test; rank 1: 1e9 s -- alone it would never fire).  The decision is collective
    (max over the ranks of elapsed - fraction * setup_time, csrc/api.hip: auto_rho_interval): both ranks fix the SAME interval at the SAME iteration,
    schedule the same rho checks, and finish with the same bits.  Before the fix rank 0 stopped calling the decision's all-reduce while
    rank 1 kept calling it: mismatched collectives -- this test then ends in the spawn timeout."""
    with tempfile.TemporaryDirectory() as tmp:
        global ITERS
        keep = ITERS
        ITERS = 130
        try:
            outs = _spawn("shm", 2, "/cosmo_test_" + uuid.uuid4().hex[:12], tmp, timeout=300,
                          extra_env={"COSMO_TEST_SHARD": "rows", "COSMO_TEST_AUTO_RHO": "0.0,1000000000.0"})
        finally:
            ITERS = keep
        for rc, o in outs:
            assert rc == 0, o[-3000:]
        z = [np.load(os.path.join(tmp, "rank%d.npz" % r)) for r in range(2)]
    assert list(z[0]["rho_interval"]) == list(z[1]["rho_interval"])
    interval, fixed_at = int(z[0]["rho_interval"][0]), int(z[0]["rho_interval"][1])
    assert interval == 25 and 0 < fixed_at <= 25                       # fired at the first test point of the plain loop: round_multiple(iter, 25) = 25
    assert int(z[0]["iter"]) == int(z[1]["iter"]) == 130 and int(z[0]["kkt"]) == int(z[1]["kkt"])
    assert len(z[0]["rho_updates"]) == len(z[1]["rho_updates"]) and np.array_equal(z[0]["rho_updates"], z[1]["rho_updates"])
    for key in ("x", "s", "y"):
        assert np.array_equal(z[0][key].view(np.int64), z[1][key].view(np.int64)), key


def _split_reference(monkeypatch, tight):
    """single-rank run of the worker's problem with the ZeroSet written as ZeroSet(1) + ZeroSet(19) (COSMO_TEST_CASE=chordal_split)"""
    W = _worker_module()
    monkeypatch.setenv("COSMO_HIP_POLAR_KLIFT", "10")
    p = W.problem()
    sets = [cj.ZeroSet(1), cj.ZeroSet(19)] + list(p["sets"][1:])
    md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], sets, W.settings(ITERS, tight))
    return cj.optimize(md), md


@pytest.mark.parametrize("world", [4, 8])
@pytest.mark.parametrize("mode", ["rows", "cones"])
def test_first_contact_readiness_four_and_eight_ranks_incl_a_rank_without_psd_cones_and_a_single_row_rank(mode, world, monkeypatch):
    """What the first real multi-GPU run (driver-side, unattended: `bench.py --gpus 8`) can meet, found on ONE GPU: 4 and 8 ranks over the
    host-staged transport, with an explicit partition in which rank 0 owns a SINGLE ROW (a ZeroSet(1)) and rank 1 owns the other simple rows -- neither owns a
    PSD cone, so their handles have no sign-iteration plan and their projections / residual partials are trivial -- while the cliques are dealt to the other
    ranks (8 ranks: one or two cliques each).  Row-sharded (csrc/rowshard.hip; the loop being sharded: src/convexset.jl:885-891 and
    src/linear_solver/kktsolver_indirect.jl:52-54): 1e-7 against the single-rank run in tight-CG mode and bit-identical ranks; clique-sharded
    (csrc/comm.hip): bit-identical to the single-rank run."""
    tight = mode == "rows"
    ref, md = _split_reference(monkeypatch, tight)
    ncones = len(md.sets)                                           # ZeroSet(1), ZeroSet(19), Nonnegatives(40), 14 cliques
    assert ncones == 17
    # ranks 0 and 1: no PSD cone; the 14 cliques over the remaining world - 2 ranks, contiguous
    rest = world - 2
    cuts = [3 + (14 * k) // rest for k in range(rest + 1)]
    bounds = [0, 1] + cuts
    assert len(bounds) == world + 1 and bounds[-1] == ncones and all(b > a for a, b in zip(bounds, bounds[1:]))
    env = {"COSMO_TEST_SHARD": mode, "COSMO_TEST_CASE": "chordal_split", "COSMO_TEST_BOUNDS": ",".join(str(b) for b in bounds)}
    if tight:
        env["COSMO_TEST_TIGHT"] = "1"
    with tempfile.TemporaryDirectory() as tmp:
        outs = _spawn("shm", world, "/cosmo_test_" + uuid.uuid4().hex[:12], tmp, timeout=420, extra_env=env)
        for rc, o in outs:
            assert rc == 0, o[-3000:]
        zs = [np.load(os.path.join(tmp, "rank%d.npz" % r)) for r in range(world)]
    rows = 0
    for r, z in enumerate(zs):
        assert int(z["nranks"]) == world and int(z["transport"]) == 2 and list(z["bounds"]) == bounds
        assert int(z["iter"]) == ref.iter == ITERS and str(z["status"]) == ref.status
        if mode == "rows":
            rows += int(z["row_hi"]) - int(z["row_lo"])
            if r == 0:
                assert int(z["row_hi"]) - int(z["row_lo"]) == 1                   # the single-row rank
            for key, val in (("x", ref.x), ("s", ref.s), ("y", ref.y)):
                assert np.max(np.abs(z[key] - val)) <= 1e-7 * max(np.max(np.abs(val)), 1e-30), (r, key)
            assert int(z["allreduces"]) >= ITERS + 1 and int(z["allreduce_elems"]) in (md.n, md.n + 2 * world)
        else:
            assert int(z["kkt"]) == ref.kkt_iters_total
            for key, val in (("x", ref.x), ("s", ref.s), ("y", ref.y)):
                assert np.array_equal(z[key].view(np.int64), val.view(np.int64)), (r, key)
    if mode == "rows":
        assert rows == md.m
    for z in zs[1:]:                                                               # identical bits on every rank
        for key in ("x", "s", "y"):
            assert np.array_equal(z[key].view(np.int64), zs[0][key].view(np.int64)), key
        assert int(z["kkt"]) == int(zs[0]["kkt"]) and float(z["obj"]) == float(zs[0]["obj"])


def test_bench_gpus_8_dry_run_prints_a_well_formed_line_with_parity_evidence():
    """`bench.py --gpus 8` as the driver launches it (torch.distributed.run, 8 ranks), all ranks on ONE GPU over the host-staged transport
    (COSMO_BENCH_TRANSPORT=shm): the row-sharded headline with `parity_ok`, eight per-rank times, the flat parity scalars the driver's parsed copy keeps
    and the summary tail.  The committed line of this command is profiles/r06_cfg5_row_sharded_8ranks_one_gpu_dryrun.json."""
    env = dict(os.environ, COSMO_BENCH_TRANSPORT="shm", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "6", "--warmup", "2", "--small", "--no-extra", "--no-cpu-baseline"]
    out = _one_json_line(subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900))
    assert out["n_gpus"] == 8 and out["steps"] == 6 and out["value"] > 0 and out["scaling"] == "strong" and "DRY RUN" in out["data"]
    cfg = out["config"]
    assert cfg["parity_ok"] is True and cfg["parity_ranks_bit_identical"] is True and 0.0 <= cfg["parity_sharded_vs_single_max_rel_dev"] <= 1e-7
    assert cfg["comm_selftest"] == "ok" and cfg["comm"]["nranks"] == 8 and len(cfg["rank_seconds"]["per_rank"]) == 8
    assert "sharded over 8 ranks" in cfg["parallelism"] and cfg["speedup_vs_single_gpu"] > 0
    assert list(out)[-1] == "summary" and out["summary"]["parity_ok"] is True


def test_row_sharded_run_with_a_user_defined_cone():
    """The AbstractConvexCone plugin surface (src/projections.jl:4-5; docs/src/literate/custom_cone.jl) on a row-sharded handle: the rank that owns the
    user's cone keeps its host callback (local cone index / row offset), the other rank has none; 300 tight-CG iterations agree with the single-rank
    run to 1e-7 and between the ranks bit for bit; the LP part reaches the documented solution x = (3, 2, 2)."""
    W = _worker_module()
    md = W.custom_cone_model(300)
    ref = cj.optimize(md)
    with tempfile.TemporaryDirectory() as tmp:
        global ITERS
        keep = ITERS
        ITERS = 300
        try:
            outs = _spawn("shm", 2, "/cosmo_test_" + uuid.uuid4().hex[:12], tmp, timeout=300, extra_env={"COSMO_TEST_SHARD": "rows", "COSMO_TEST_CASE": "custom"})
        finally:
            ITERS = keep
        for rc, o in outs:
            assert rc == 0, o[-3000:]
        z = [np.load(os.path.join(tmp, "rank%d.npz" % r)) for r in range(2)]
    assert int(z[0]["row_hi"]) == int(z[1]["row_lo"]) and int(z[0]["row_lo"]) == 0 and int(z[0]["row_hi"]) > 0        # both ranks own rows
    for key, val in (("x", ref.x), ("s", ref.s), ("y", ref.y)):
        assert np.array_equal(z[0][key].view(np.int64), z[1][key].view(np.int64)), key
        assert np.max(np.abs(z[0][key] - val)) <= 1e-7 * max(1.0, float(np.max(np.abs(val)))), key
    np.testing.assert_allclose(z[0]["x"][:3], [3.0, 2.0, 2.0], atol=1e-3)


def test_row_sharded_run_with_the_reduced_minres_solver(monkeypatch):
    """IndirectReducedKKTSolver with solver_type = :MINRES (src/linear_solver/kktsolver_indirect.jl:3-88) on a row-sharded handle: MINRES runs
    replicated on the split reduced operator (the multi-nonzero rows of the WHOLE A + the diagonal of the singleton rows), the right-hand side comes
    from the all-reduce the CG path uses.  60 tight iterations: ranks bit-identical, 1e-7 against the single-rank MINRES run (whose operator is the
    unsplit one: a re-association)."""
    monkeypatch.setenv("COSMO_TEST_KKT", "minres_reduced")
    monkeypatch.setenv("COSMO_HIP_POLAR_KLIFT", "10")
    W = _worker_module()
    p = W.problem()
    md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], W.settings(ITERS))
    ref = cj.optimize(md)
    with tempfile.TemporaryDirectory() as tmp:
        outs = _spawn("shm", 2, "/cosmo_test_" + uuid.uuid4().hex[:12], tmp, extra_env={"COSMO_TEST_SHARD": "rows", "COSMO_TEST_KKT": "minres_reduced"})
        for rc, o in outs:
            assert rc == 0, o[-3000:]
        z = [np.load(os.path.join(tmp, "rank%d.npz" % r)) for r in range(2)]
    assert int(z[0]["iter"]) == int(z[1]["iter"]) == ref.iter == ITERS and str(z[0]["mode"]) == "rows"
    for key, val in (("x", ref.x), ("s", ref.s), ("y", ref.y)):
        assert np.array_equal(z[0][key].view(np.int64), z[1][key].view(np.int64)), key
        assert np.max(np.abs(z[0][key] - val)) <= 1e-7 * max(1.0, float(np.max(np.abs(val)))), key


def test_row_sharded_run_with_the_jacobi_preconditioned_cg(monkeypatch):
    """The opt-in Jacobi-preconditioned CG (kkt_kind CG_JACOBI, csrc/cg_fold.hip) on a row-sharded handle: the assembled operator, its diagonal and
    the Krylov vectors are replicated, so the ranks stay bit-identical; against the single-rank run of the same solver 1e-7 after 60 tight iterations
    (the sharded right-hand side is a sum of per-rank partial products: a re-association)."""
    monkeypatch.setenv("COSMO_TEST_KKT", "cg_jacobi")
    monkeypatch.setenv("COSMO_HIP_POLAR_KLIFT", "10")
    W = _worker_module()
    p = W.problem()
    md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], W.settings(ITERS))
    ref = cj.optimize(md)
    with tempfile.TemporaryDirectory() as tmp:
        outs = _spawn("shm", 2, "/cosmo_test_" + uuid.uuid4().hex[:12], tmp, extra_env={"COSMO_TEST_SHARD": "rows", "COSMO_TEST_KKT": "cg_jacobi"})
        for rc, o in outs:
            assert rc == 0, o[-3000:]
        z = [np.load(os.path.join(tmp, "rank%d.npz" % r)) for r in range(2)]
    assert int(z[0]["iter"]) == int(z[1]["iter"]) == ref.iter == ITERS and str(z[0]["mode"]) == "rows"
    assert int(z[0]["kkt"]) == int(z[1]["kkt"]) and abs(int(z[0]["kkt"]) - ref.kkt_iters_total) <= 0.05 * ref.kkt_iters_total + 10
    for key, val in (("x", ref.x), ("s", ref.s), ("y", ref.y)):
        assert np.array_equal(z[0][key].view(np.int64), z[1][key].view(np.int64)), key
        assert np.max(np.abs(z[0][key] - val)) <= 1e-7 * max(1.0, float(np.max(np.abs(val)))), key


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _one_json_line(p):
    import json
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                               # exactly one JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("workload,shard", [("cfg5", "rows"), ("cfg5", "cones"), ("cfg3", "rows")])
def test_bench_multi_rank_entry_point_dry_run(workload, shard):
    """`bench.py --gpus 2` launched by torch.distributed.run (one rank per process), on ONE GPU: the dry-run transport
    (COSMO_BENCH_TRANSPORT=shm: gloo barriers, host-staged collectives) exercises the rank bookkeeping, the sharded cfg5 / cfg3
    workloads, the MAX-over-ranks timing and the one JSON line of rank 0.  With real GPUs the same code runs over RCCL."""
    env = dict(os.environ, COSMO_BENCH_TRANSPORT="shm", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "3", "--small", "--workload", workload, "--shard", shard]
    out = _one_json_line(subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420))
    assert out["n_gpus"] == 2 and out["steps"] == 8 and out["warmup"] == 3 and out["value"] > 0 and out["higher_is_better"] is True
    assert "DRY RUN" in out["data"]
    if workload == "cfg5":
        assert out["scaling"] == "strong"
        comm = out["config"]["comm"]
        assert comm["nranks"] == 2 and comm["transport"] == 2 and comm["mode"] == shard and comm["collectives"] >= 8 + 3
        n = 6000
        if shard == "rows":                                              # one all-reduce of n doubles per iteration (+ the checks' n + 4)
            assert 8 * n <= comm["bytes_per_iteration"] <= 8 * (n + 4) * 1.5 and 1.0 <= comm["collectives_per_iteration"] <= 1.5
            info = out["config"]["row_shard"]
            assert 0 <= info["row_lo"] < info["row_hi"] < info["m_global"]
        else:                                                            # the projected PSD slices of s: far more than an n-vector
            assert comm["bytes_per_iteration"] > 8 * 4 * n and comm["collectives_per_iteration"] == 1.0
        assert out["config"]["speedup_vs_single_gpu"] > 0 and "sharded over 2 ranks" in out["config"]["parallelism"]
        _assert_parity_evidence(out["config"], comm, 1e-7)
        sh = out["config"]["shardable_share_of_single_gpu_iteration"]
        assert 0 < sh["projections"] <= sh["projections_and_row_kernels"] < 1 and 1 < sh["predicted_speedup_bound"] < 2
    else:
        assert out["scaling"] == "strong" and "sharded over 2 rank" in out["config"]["parallelism"]
        par = out["config"]["parity"]
        assert par["ok"] is True and par["sharded_vs_single_max_rel_dev"] == 0.0 and par["problems_compared"] == 32     # 16 from each rank's shard
        rs = out["config"]["rank_seconds"]
        assert len(rs["per_rank"]) == 2 and 0 < rs["min"] <= rs["max"]


def _assert_parity_evidence(cfg, comm, tol):
    """What VERDICT r03 item 1 asks of every N > 1 line: (a) the known-answer all-reduce right after comm_init, (b) the sharded iterates against
    rank 0's unsharded run of the same problem, (d) per-rank times, (e) transport and RCCL version."""
    assert comm["selftest"] == "ok" and comm["exact_sum_mismatches"] == 0 and comm["fractional_sum_outside_bound"] == 0
    assert comm["result_bits_identical_on_all_ranks"] is True and comm["count"] > 0
    assert "host-staged" in comm["transport_name"] and comm["rccl_version"] is None            # the dry-run transport (RCCL: "rccl" and a version string)
    par = cfg["parity"]
    assert par["ok"] is True and 0.0 <= par["sharded_vs_single_max_rel_dev"] <= tol and par["ranks_bit_identical"] is True
    assert par["krylov_iterations"]["single"] > 0 and par["krylov_iterations"]["sharded"] > 0
    rs = cfg["rank_seconds"]
    assert len(rs["per_rank"]) == 2 and 0 < rs["min"] <= rs["max"]


def test_a_corrupted_exchange_turns_the_bench_lines_parity_evidence_red():
    """COSMO_HIP_COMM_CORRUPT_RANK=1 (test hook in csrc/comm.hip) makes rank 1 contribute 1.001 x its vector to every all-reduce: the known-answer
    check and the sharded-vs-single deviation of the SAME bench line must both say so -- a throughput number with a wrong exchange cannot pass."""
    env = dict(os.environ, COSMO_BENCH_TRANSPORT="shm", MASTER_ADDR="127.0.0.1", COSMO_HIP_COMM_CORRUPT_RANK="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--small", "--workload", "cfg5", "--shard", "rows"]
    out = _one_json_line(subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420))
    comm, par = out["config"]["comm"], out["config"]["parity"]
    assert comm["selftest"].startswith("FAILED") and comm["exact_sum_mismatches"] > 0
    assert par["ok"] is False and par["sharded_vs_single_max_rel_dev"] > 1e-5
    # ... and so do the FLAT top-level copies the driver's parsed line keeps (round 5), and the digest at the end of the line
    cfg = out["config"]
    assert cfg["parity_ok"] is False and cfg["parity_sharded_vs_single_max_rel_dev"] > 1e-5 and cfg["comm_selftest"].startswith("FAILED")
    assert out["summary"]["parity_ok"] is False and list(out)[-1] == "summary"


def test_bench_gpus_2_without_a_launcher_is_a_two_rank_run_of_the_same_headline_workload():
    """The driver starts the bench as a plain `python bench.py --gpus N ...`: that command alone must produce a valid N-rank line (it
    re-launches itself under torch.distributed.run), and the headline must be the SAME workload at every N -- since round 5 config 5, the workload
    north_star's targets are stated on: ONE problem, unsharded at N = 1, row-sharded at N > 1, `scaling: strong` (the 1 -> 8 curve
    of weak-scaled cfg2 replicas read 8x by construction).  The parity evidence sits at the TOP level of the line; the batch that shards is
    extra.cfg3_sharded; the two configurations that do not shard are explicit replicas, one replica's rate, never aggregated."""
    common = ["--steps", "6", "--warmup", "2", "--small", "--no-cpu-baseline", "--no-float32"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    one = _one_json_line(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-extra"] + common, env=env, cwd=ROOT,
                                        capture_output=True, text=True, timeout=420))
    two = _one_json_line(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, env=dict(env, COSMO_BENCH_TRANSPORT="shm"),
                                        cwd=ROOT, capture_output=True, text=True, timeout=600))
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["config"]["workload"] == two["config"]["workload"] and one["config"]["workload"].startswith("cfg5")
    assert one["metric"] == two["metric"] and one["scaling"] == two["scaling"] == "strong" and one["steps"] == two["steps"] == 6
    assert two["config"]["launch"] == "self-launched torch.distributed.run" and one["config"]["launch"] == "single process"
    assert one["config"]["parallelism"] == "single GPU" and "sharded over 2 ranks" in two["config"]["parallelism"] and two["value"] > 0 and "DRY RUN" in two["data"]
    assert one["roofline"]["bound"] in ("hbm", "mfma") and two["roofline"]["bound"] in ("hbm", "mfma")
    c5 = two["config"]
    assert c5["comm"]["nranks"] == 2 and c5["comm"]["mode"] == "rows" and c5["single_gpu_same_workload"] > 0 and c5["speedup_vs_single_gpu"] > 0
    _assert_parity_evidence(c5, c5["comm"], 1e-7)
    # the flat copies (what the driver's parsed line keeps) agree with the nested objects
    assert c5["parity_ok"] is True and c5["parity_sharded_vs_single_max_rel_dev"] == c5["parity"]["sharded_vs_single_max_rel_dev"] <= 1e-7
    assert c5["parity_ranks_bit_identical"] is True and c5["comm_selftest"] == "ok" and "host-staged" in c5["comm_transport"]
    ex = two["extra"]
    assert set(ex) == {"cfg3_sharded", "cfg2_replicas", "cfg4_replicas"}
    for key in ex:
        assert "error" not in ex[key], ex[key]
        assert ex[key]["n_gpus"] == 2 and ex[key]["value"] > 0
    assert ex["cfg3_sharded"]["scaling"] == "strong"
    assert ex["cfg3_sharded"]["config"]["parity"]["ok"] is True and ex["cfg3_sharded"]["config"]["parity"]["sharded_vs_single_max_rel_dev"] == 0.0
    for key in ("cfg2_replicas", "cfg4_replicas"):
        assert ex[key]["scaling"] == "replicas" and ex[key]["replicas"] == 2
        assert abs(ex[key]["value"] * ex[key]["ms_per_step"] / 1e3 - 1.0) < 1e-3      # ONE replica's rate = steps / max-over-ranks time: not multiplied by N
    # the digest at the very end of the line (the driver keeps the tail): every workload's rate
    assert list(two)[-1] == "summary" and set(two["summary"]) >= {"cfg5", "cfg3_sharded", "cfg2_replicas", "cfg4_replicas", "parity_ok", "speedup_vs_single_gpu"}
    assert two["summary"]["cfg5"]["value"] == two["value"] and two["summary"]["parity_ok"] is True


def test_bench_reports_the_headline_even_if_a_sharded_extra_never_finishes():
    """The sharded extras are the only part of bench.py with data-path collectives; if one of them hung (a rank failing while the others
    wait in a collective) no line at all would come out of an N-GPU run.  A watchdog prints the headline -- measured before the extras
    start -- with what has been collected and ends every rank.  Here the deadline is zero, so it fires while the first extra is being set up."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env.update(COSMO_BENCH_TRANSPORT="shm", COSMO_BENCH_EXTRA_TIMEOUT="0.05", COSMO_BENCH_EXTRA_GRACE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--small", "--no-cpu-baseline", "--no-float32"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    two = _one_json_line(r)
    assert two["n_gpus"] == 2 and two["value"] > 0 and two["config"]["workload"].startswith("cfg5") and two["scaling"] == "strong"
    assert set(two["extra"]) == {"cfg3_sharded", "cfg2_replicas", "cfg4_replicas"}
    assert any("error" in v for v in two["extra"].values())
