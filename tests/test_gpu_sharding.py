"""GPU tests of the clique-sharded projection on ONE device: ownership ranges + slice merge reproduce the full
projection bit for bit; the RCCL path is exercised with a single-rank communicator."""
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


def test_single_rank_communicator_rccl_path():
    prob = cj.problems.chordal_sdp(ncliques=8, dmin=4, dmax=30, sep_min=1, sep_max=3, n_total=600, n_zero=5, n_nonneg=10)
    st = cj.Settings(max_iter=100, eps_abs=0, eps_rel=0)
    ref_model = cj.Model(); ref_model.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], st)
    ref = cj.optimize(ref_model)
    model = cj.Model(); model.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], st)
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
