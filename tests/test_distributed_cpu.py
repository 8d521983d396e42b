"""CPU tests of the N > 1 path (gloo, world_size 2): batches of independent problems shard over ranks with no data-path
collective; results are exchanged once at the end.  The device solver is replaced by an injected shard solver (the oracle)
because this container has no GPU -- what is tested is the sharding / gathering logic the GPU ranks run."""
import os

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import cosmo_jl_amd as cj
from oracle import cosmo_oracle as O
from tests import util


def _oracle_shard_solver(models, device):
    out = []
    for md in models:
        r = O.solve(md.P, md.q, md.A, md.b, util.oracle_cones(md.sets), O.Settings(kkt_solver="cg", max_iter=400))
        out.append(cj.model.Result(x=r.x, y=r.y, s=r.s, obj_val=r.obj_val, iter=r.iter, status=r.status,
                                   info=cj.model.ResultInfo(r.r_prim, r.r_dual, r.max_norm_prim, r.max_norm_dual, r.rho_updates),
                                   times=cj.model.ResultTimes()))
    return out


def _models():
    probs = [cj.problems.socp(n=20, m=30, ncones=5, nnz=120, seed=7 + k) for k in range(5)]
    out = []
    for p in probs:
        md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], cj.Settings())
        out.append(md)
    return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = cj.optimize_batch(_models(), dist=dist, solve_shard=_oracle_shard_solver)
    q.put((rank, [(r.status, r.iter, float(r.obj_val)) for r in res]))
    dist.barrier()
    dist.destroy_process_group()


def test_batch_shards_over_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    got = dict(q.get(timeout=240) for _ in range(world))
    for p in procs: p.join(timeout=60)
    serial = [(r.status, r.iter, float(r.obj_val)) for r in _oracle_shard_solver(_models(), 0)]
    assert got[0] == got[1] == serial                       # every rank ends with all results, in problem order


def test_shard_range_and_cone_balance():
    for n_items in (0, 1, 7, 1024):
        for world in (1, 2, 3, 8):
            rs = [cj.shard_range(n_items, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n_items
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1
    rng = np.random.default_rng(5)
    dims = rng.integers(20, 201, size=400)
    own = cj.balance_cones(dims, 8)
    assert sorted(i for o in own for i in o) == list(range(400))
    loads = [sum(int(dims[i]) ** 3 for i in o) for o in own]
    assert max(loads) <= 1.05 * (sum(loads) / 8) + 200 ** 3    # LPT bound: within one largest item of the mean


def test_contiguous_cone_partition():
    rng = np.random.default_rng(5)
    for trial in range(20):
        ncones = int(rng.integers(1, 60)); world = int(rng.integers(1, 9))
        costs = [int(c) for c in rng.integers(0, 50, size=ncones) ** 3]
        b = cj.partition_cones_contiguous(costs, world)
        assert len(b) == world + 1 and b[0] == 0 and b[-1] == ncones and all(b[i] <= b[i + 1] for i in range(world))
        loads = [sum(costs[b[i]:b[i + 1]]) for i in range(world)]
        # optimal bottleneck: no contiguous partition into `world` parts can beat it (check against brute-force lower bounds)
        assert max(loads) >= max(max(costs), -(-sum(costs) // world)) and max(loads) <= max(costs) + sum(costs) // world + 1
    sets = [cj.ZeroSet(3), cj.Nonnegatives(4), cj.SecondOrderCone(5), cj.PsdConeTriangle(10), cj.PsdCone(9), cj.PsdConeTriangle(1)]
    assert cj.cone_costs(sets) == [0, 0, 5, 64, 27, 0]


def _bench_timing_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    dist.barrier()
    mine = 0.5 + 0.25 * rank                                 # rank 1 is the slow one
    job = bench.max_over_ranks(mine, dist, "cpu")
    q.put((rank, job, bench.whole_job_value(world, 10, job)))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_timing_rule_max_over_ranks_gloo():
    """bench.py's contract: K steps per rank bracketed by barriers, the job time is the MAX over ranks, `value` is the whole-job
    aggregate (replicas: world * K / time).  The two helpers are the ones bench.py itself calls (there with device = "cuda")."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_bench_timing_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    got = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs: p.join(timeout=60)
    assert [g[1] for g in got] == [0.75, 0.75]
    assert all(abs(g[2] - 2 * 10 / 0.75) < 1e-12 for g in got)


# ---- round 3: the host-side pieces of the row-sharded run and of bench.py's launcher / CPU-baseline plumbing (no GPU involved) ----
def test_row_shard_partition_tiles_the_cone_list_and_balances_rows_and_cubes():
    """cj.model.row_shard_costs + partition_cones_contiguous: what every rank of a row-sharded run computes identically (csrc/rowshard.hip takes
    the boundaries).  Contiguous, covers all cones, every rank non-empty on the BASELINE config 5 structure, bottleneck within 25 % of the mean."""
    p = cj.problems.chordal_sdp(ncliques=60, dmin=20, dmax=120, n_total=8000, n_zero=200, n_nonneg=900, seed=9)
    costs = cj.model.row_shard_costs(p["sets"])
    assert len(costs) == len(p["sets"]) and costs[0] == 8 * 200 and costs[1] == 8 * 900       # simple cones cost their rows only
    for world in (1, 2, 3, 8):
        b = cj.partition_cones_contiguous(costs, world)
        assert b[0] == 0 and b[-1] == len(costs) and all(b[r] <= b[r + 1] for r in range(world))
        loads = [sum(costs[b[r]:b[r + 1]]) for r in range(world)]
        assert all(l > 0 for l in loads)
        assert max(loads) <= 1.25 * sum(costs) / world + max(costs)
    # exponential / power cones shard with the rows (cost 64 + rows) but not with the cones (clique mode projects them everywhere: cost 0)
    sets = [cj.ZeroSet(3), cj.ExponentialCone(), cj.PowerCone(0.3), cj.SecondOrderCone(5)]
    assert cj.cone_costs(sets) == [0, 0, 0, 5] and cj.model.row_shard_costs(sets) == [24, 64 + 24, 64 + 24, 5 + 40]


def test_bench_cpu_leg_plumbing():
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
        B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
    finally:
        sys.argv = argv
    assert B._cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11} and B._cpulist("") == set()
    cpus, where = B.numa_node0_physical_cores(cap=4)
    assert 1 <= len(cpus) <= 4 and set(cpus) <= set(os.sched_getaffinity(0)) and "physical cores" in where
    # the compiled loop as the bench times it (a small SOCP: SecondOrderCone projections in C)
    p = cj.problems.socp(n=40, m=60, ncones=6, nnz=300, seed=3)
    r = B.compiled_cpu_rate(p, 30, 1)
    assert r["iters"] == 30 and r["rate"] > 0 and r["secs"] > 0 and r["cg"] > 0
    assert B.whole_job_value(4, 10, 2.0) == 20.0


# ---- pure-Python parts of the N > 1 bench line (bench.py cannot run on this GPU-less host; its bookkeeping can) ----------------------------
def _bench_module():
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_cfg3_scaling_model_says_chain_limited_when_one_problem_dominates():
    """The batch step is t_k * max(longest chain on a rank, the rank's work / 256 CU slots): with BASELINE config 3's measured distribution (a few
    problems with thousands of Krylov iterations, median 50) every N is chain limited and the predicted sharded speed-up is 1.0; a flat distribution
    with more problems than slots is work limited and scales."""
    B = _bench_module()
    rng = np.random.default_rng(0)
    K = rng.integers(8, 140, 1024).astype(float)
    K[[133, 2, 613, 606]] = [3304, 2929, 2852, 1844]
    m = B.cfg3_scaling_model(cj, K, elapsed=3304 * 7.0e-6, world=1)
    assert m["longest_chain_krylov_iterations"] == 3304 and abs(m["seconds_per_krylov_iteration"] - 7.0e-6) < 1e-12
    assert all(m["predicted"][str(N)]["chain_limited"] and m["predicted"][str(N)]["speedup_bound"] == 1.0 for N in (1, 2, 4, 8))
    assert 0.05 < m["cu_slot_utilisation_this_run"] < 0.2
    flat = np.full(4096, 100.0)                                # 4096 equal problems on 256 slots: 16 waves of work per rank at N = 1
    f = B.cfg3_scaling_model(cj, flat, elapsed=1600 * 7.0e-6, world=1)
    assert not f["predicted"]["1"]["chain_limited"] and f["predicted"]["8"]["speedup_bound"] == pytest.approx(8.0)
    # measured on several ranks: t_k comes from the slowest rank's longest chain
    m2 = B.cfg3_scaling_model(cj, K, elapsed=3304 * 7.0e-6, world=2)
    assert abs(m2["seconds_per_krylov_iteration"] - 7.0e-6) < 1e-12


class _FakeCtx:
    def __init__(self, results):
        self.results = results

    def all_gather(self, obj):
        return self.results


class _FakeHandle:
    def __init__(self, out=None, err=None):
        self.out, self.err = out, err

    def comm_allreduce_check(self, count):
        if self.err:
            raise RuntimeError(self.err)
        return self.out


def test_comm_known_answer_check_combines_the_ranks_verdicts():
    B = _bench_module()
    good = dict(exact_mismatches=0, inexact_outside_bound=0, hash=1234, transport=1, nranks=2, rccl_version_code=22606, count=100)
    ok = B.comm_known_answer_check(_FakeCtx([(good, None), (dict(good), None)]), _FakeHandle(good), 100)
    assert ok["selftest"] == "ok" and ok["rccl_version"] == "2.26.6" and ok["transport_name"] == "rccl" and ok["result_bits_identical_on_all_ranks"] is True
    other_bits = dict(good, hash=999)
    bad = B.comm_known_answer_check(_FakeCtx([(good, None), (other_bits, None)]), _FakeHandle(good), 100)
    assert bad["selftest"].startswith("FAILED") and bad["result_bits_identical_on_all_ranks"] is False
    wrong_sum = dict(good, exact_mismatches=7)
    bad = B.comm_known_answer_check(_FakeCtx([(good, None), (wrong_sum, None)]), _FakeHandle(good), 100)
    assert bad["selftest"].startswith("FAILED") and bad["exact_sum_mismatches"] == 7
    err = B.comm_known_answer_check(_FakeCtx([(good, None), (None, "RuntimeError: ncclAllReduce failed")]), _FakeHandle(good), 100)
    assert err["selftest"].startswith("FAILED") and "ncclAllReduce" in err["selftest"]
