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


def _spawn(transport, world, rdv, tmp, timeout=240):
    env = dict(os.environ, COSMO_HIP_POLAR_KLIFT="10")            # pin the sign-iteration schedule (it is pinned in sharded runs anyway)
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


def _single_rank_reference(monkeypatch):
    import importlib.util
    spec = importlib.util.spec_from_file_location("shard_worker", WORKER)
    W = importlib.util.module_from_spec(spec); spec.loader.exec_module(W)
    monkeypatch.setenv("COSMO_HIP_POLAR_KLIFT", "10")
    p = W.problem()
    md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], W.settings(ITERS))
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


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("workload", ["cfg5", "cfg3"])
def test_bench_multi_rank_entry_point_dry_run(workload):
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one rank per process), on ONE GPU: the dry-run
    transport (COSMO_BENCH_TRANSPORT=shm: gloo barriers, host-staged clique exchange) exercises the rank bookkeeping, the sharded cfg5 /
    cfg3 workloads, the MAX-over-ranks timing and the one JSON line of rank 0.  With real GPUs the same code runs over RCCL."""
    env = dict(os.environ, COSMO_BENCH_TRANSPORT="shm", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "3", "--small", "--workload", workload]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                               # exactly one JSON line, from rank 0
    import json
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 8 and out["warmup"] == 3 and out["value"] > 0 and out["higher_is_better"] is True
    assert "DRY RUN" in out["data"]
    if workload == "cfg5":
        assert out["scaling"] == "strong"
        comm = out["config"]["comm"]
        assert comm["nranks"] == 2 and comm["transport"] == 2 and comm["exchanges"] >= 8 + 3
        assert out["config"]["speedup_vs_single_gpu"] > 0 and "sharded over 2 ranks" in out["config"]["parallelism"]
    else:
        assert out["scaling"] == "weak" or out["scaling"] == "strong"
        assert "sharded over 2 rank" in out["config"]["parallelism"]
