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
