"""NaN propagation of the residual norms.  Julia's `norm(x, Inf)` returns NaN when x holds one
(/root/reference/src/residuals.jl:30-53), which makes `has_converged` false (`:98-117`): a run whose residual vector contains a NaN can end
`Max_iter_reached`, never `Solved`.  The device's per-thread accumulation always kept a NaN; its cross-lane steps (`wave_max`, `block_max`:
csrc/device_utils.h) dropped one held by the upper half-wave / a later wave until round 5.

The plant that keeps a NaN CONFINED TO ONE ROW through a whole run: an EMPTY row i of A (no stored entry, so no sparse product ever touches
it -- the reference's CSC kernels skip structural zeros too) with b_i = NaN.  Then r_prim_i = (Ax)_i + s_i - b_i is NaN at every check while all
other rows, x, and the dual residual stay finite and converge as if row i did not exist.  The row index chooses the lane / wave that owns it."""
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
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


nan_row_qp = util.nan_row_qp


def _settings(max_iter=400, **kw):
    return cj.Settings(max_iter=max_iter, scaling=0, kkt_solver=cj.with_options(cj.CGIndirectKKTSolver, tol_constant=1e-10, tol_exponent=0.0), **kw)


def _solve(p, dtype=np.float64, **kw):
    md = cj.Model(dtype=dtype)
    md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], _settings(**kw))
    return cj.optimize(md)


# rows chosen by their owner in the CSR-stream tile of k_chk_prim (thread = row - first row of the tile, 256-thread workgroups): lower half of
# wave 0; upper half of wave 0; upper half of wave 1; a later tile
@pytest.mark.parametrize("nan_row", [5, 37, 100, 163, 1900])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_single_problem_nan_in_one_residual_row_never_reads_solved(nan_row, dtype):
    n, m = 60, 2000
    clean = _solve(nan_row_qp(3, n, m, None), dtype)
    assert clean.status == "Solved" and clean.iter < 400                   # the control: without the plant the run converges well before max_iter
    p = nan_row_qp(3, n, m, nan_row)
    r = _solve(p, dtype)
    assert r.status == "Max_iter_reached" and r.iter == 400
    assert np.isnan(r.info.r_prim) and np.isnan(r.info.max_norm_prim)     # norm(., Inf) of a vector holding a NaN
    assert np.isfinite(r.info.r_dual) and np.isfinite(r.info.max_norm_dual) and np.all(np.isfinite(r.x))
    if dtype is np.float64:
        # the oracle (= the reference's loop) on the same instance: same status, same count, NaN in the same places, finite parts equal
        st = O.Settings(scaling=0, kkt_solver="cg", tol_constant=1e-10, tol_exponent=0.0, max_iter=400)
        ws = O.Workspace(p["P"], p["q"], p["A"], p["b"], util.oracle_cones(p["sets"]), st)
        ref = ws.optimize()
        assert ref.status == "Max_iter_reached" and ref.iter == 400 and np.isnan(ref.r_prim) and np.isfinite(ref.r_dual)
        assert abs(r.info.r_dual - ref.r_dual) <= 1e-7 * max(ref.max_norm_dual, 1.0)
        assert np.max(np.abs(r.x - ref.x)) <= 1e-7 * max(np.max(np.abs(ref.x)), 1.0)
        # a NaN residual ratio leaves rho alone (Julia's min(max(NaN, lo), hi) is NaN, both comparisons of parameters.jl:67 fail)
        assert len(r.info.rho_updates) == len(ref.rho_updates) == 1


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_residuals_entry_point_keeps_a_nan_of_any_lane(dtype):
    """cosmo_hip_residuals (recover_mu! + calculate_result_info!, residuals.jl:30-96): b_i = NaN in a row with entries poisons r_prim and
    max_norm_prim only; q_j = NaN poisons r_dual, max_norm_dual and the cost only.  Every row / column position of a 256-thread tile is tried."""
    n, m = 300, 700
    rng = np.random.default_rng(8)
    A = sp.random(m, n, density=0.02, random_state=rng, format="csc", data_rvs=rng.standard_normal)
    P = sp.identity(n, format="csc")
    x0 = rng.standard_normal(n); s0 = np.abs(rng.standard_normal(m)); mu0 = rng.standard_normal(m)
    for pos in (0, 31, 32, 63, 64, 97, 200, 255, 256, 299):
        for which in ("b", "q"):
            b = rng.standard_normal(m); q = rng.standard_normal(n)
            (b if which == "b" else q)[pos] = np.nan
            h = cj.Handle(0, dtype=dtype)
            h.set_problem(P, q, A, b)
            h.set_cones([F.NONNEG], [m], None, None)
            h.set_params(h.default_params())
            h.set_iterates(x0, s0, mu0)
            out = h.residuals()
            if which == "b":
                assert np.isnan(out[0]) and np.isnan(out[2]) and np.isfinite(out[1]) and np.isfinite(out[3]) and np.isfinite(out[4]), (pos, out)
            else:
                assert np.isnan(out[1]) and np.isnan(out[3]) and np.isnan(out[4]) and np.isfinite(out[0]) and np.isfinite(out[2]), (pos, out)


@pytest.mark.parametrize("variant", ["reg", "lds", "stream"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_batch_nan_in_one_residual_row_of_one_problem(variant, dtype, monkeypatch):
    """The persistent batch kernels take the same decisions per problem (csrc/batch.hip: bmax): the poisoned problem ends Max_iter_reached with a
    NaN r_prim, its neighbours are Solved as if it were not there.  Element i of an iterate lives in thread i mod 512 (register kernel)."""
    if variant == "lds":
        monkeypatch.setenv("COSMO_HIP_BATCH_REG", "0")
    elif variant == "stream":
        monkeypatch.setenv("COSMO_HIP_BATCH_LDS", "0")
    n, m = 40, 700
    rows = [None, 100, None, 37, 613, None]                                   # 100: wave 1 upper half; 37: wave 0 upper half; 613 = 512 + 101
    models = []
    for k, row in enumerate(rows):
        p = nan_row_qp(20 + k, n, m, row)
        md = cj.Model(dtype=dtype); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], _settings(max_iter=300))
        models.append(md)
    res = cj.optimize_batch(models)
    for row, r in zip(rows, res):
        if row is None:
            assert r.status == "Solved" and r.iter < 300 and np.isfinite(r.info.r_prim)
        else:
            assert r.status == "Max_iter_reached" and r.iter == 300, (row, r.status, r.iter)
            assert np.isnan(r.info.r_prim) and np.isfinite(r.info.r_dual)


def test_row_sharded_nan_in_one_residual_row(monkeypatch):
    """Two ranks (host-staged transport, one GPU): the NaN row belongs to rank 1, whose primal norm travels to rank 0 in the all-reduced slots
    behind A'mu (k_rs_pack_prim / k_chk_dual_rs); both ranks must report Max_iter_reached with a NaN r_prim, bit-identically."""
    worker = os.path.join(ROOT, "tests", "shard_worker.py")
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, COSMO_TEST_SHARD="rows", COSMO_TEST_CASE="nanrow")
        rdv = "/cosmo_test_" + uuid.uuid4().hex[:12]
        procs = [subprocess.Popen([sys.executable, worker, "shm", str(r), "2", rdv, os.path.join(tmp, "rank%d.npz" % r), "300"], env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
        outs = [p.communicate(timeout=240) for p in procs]
        assert all(p.returncode == 0 for p in procs), outs
        r0, r1 = (np.load(os.path.join(tmp, "rank%d.npz" % r)) for r in range(2))
        for r in (r0, r1):
            assert str(r["status"]) == "Max_iter_reached" and int(r["iter"]) == 300 and np.isnan(float(r["r_prim"])) and np.isfinite(float(r["r_dual"]))
            assert int(r["row_hi"]) - int(r["row_lo"]) > 0
        assert int(r1["row_lo"]) <= 1200 + 100 < int(r1["row_hi"])            # the planted row is rank 1's
        assert np.array_equal(r0["x"], r1["x"]) and float(r0["r_dual"]) == float(r1["r_dual"])
