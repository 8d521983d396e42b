"""The single-launch persistent CG (csrc/cg_persist.hip) reproduces the multi-kernel Krylov loop BIT FOR BIT: same row sums, same
reduction trees, same iteration counts -- on a split operator (decomposed SDP), on an unsplit one (QP with Box / SOC rows) and
through the fine-grained AbstractKKTSolver entry point (reference: src/linear_solver/kktsolver_indirect.jl:57-70)."""
import numpy as np
import pytest

import cosmo_jl_amd as cj
from oracle import cosmo_oracle as O
from tests import util

pytestmark = pytest.mark.gpu


def _solve(monkeypatch, persist, prob, iters, fuse="1", **st_kw):
    monkeypatch.setenv("COSMO_HIP_CG_PERSIST", persist)
    monkeypatch.setenv("COSMO_HIP_CG_FUSE_DIR", fuse)
    monkeypatch.setenv("COSMO_HIP_OP_FOLD", "0")       # these tests compare the kernels of the split / unsplit operator (no assembled M, csrc/cg_fold.hip)
    monkeypatch.setenv("COSMO_HIP_POLAR_KLIFT", "10")
    st = cj.Settings(max_iter=iters, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9, **st_kw)
    md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], st)
    r = cj.optimize(md)
    rhs = np.random.default_rng(5).standard_normal(md.n + md.m)
    sol, its = md.handle.kkt_solve(rhs)
    return r, sol, its, md.handle.cg_persist_stats()


PROBLEMS = {
    "chordal_sdp_split_operator": lambda: cj.problems.chordal_sdp(ncliques=12, dmin=4, dmax=70, sep_min=1, sep_max=3, n_total=2500, n_zero=40, n_nonneg=80),
    "box_qp": lambda: cj.problems.sparse_box_qp(n=3000, m=6000, nnz=50000, seed=9),
    "mixed_cones_qp": lambda: util.random_qp(np.random.default_rng(21), 400, 20, 150, 120, soc_dims=(7, 12, 30), density=0.03, p_shift=0.5),
}


@pytest.mark.parametrize("name", sorted(PROBLEMS))
def test_persistent_cg_is_bit_identical_to_the_multi_kernel_loop(name, monkeypatch):
    prob = PROBLEMS[name]()
    r1, s1, k1, st1 = _solve(monkeypatch, "1", prob, 80)
    r0, s0, k0, st0 = _solve(monkeypatch, "0", prob, 80)
    assert st1["enabled"] == 1 and st1["launches"] >= 80 and st1["fallbacks"] == 0, st1      # the persistent kernel really ran
    assert st0["enabled"] == 0 and st0["launches"] == 0
    assert r1.iter == r0.iter == 80 and r1.kkt_iters_total == r0.kkt_iters_total and k1 == k0
    for a, b in ((r1.x, r0.x), (r1.s, r0.s), (r1.y, r0.y), (s1, s0)):
        assert np.array_equal(a.view(np.int64), b.view(np.int64))
    assert r1.obj_val == r0.obj_val and r1.info.r_prim == r0.info.r_prim and r1.info.r_dual == r0.info.r_dual
    assert r1.info.rho_updates == r0.info.rho_updates


def test_persistent_cg_default_solve_matches_the_oracle(monkeypatch):
    """Default (loose, 1/k^1.5) CG tolerance, termination by the residual test: same status / iteration count as the oracle."""
    monkeypatch.setenv("COSMO_HIP_CG_PERSIST", "1")
    prob = cj.problems.sparse_box_qp(n=2000, m=4000, nnz=30000, seed=3)
    md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], cj.Settings())
    r = cj.optimize(md)
    ref = O.solve(prob["P"], prob["q"], prob["A"], prob["b"], util.oracle_cones(prob["sets"]), O.Settings(kkt_solver="cg"))
    assert md.handle.cg_persist_stats()["launches"] > 0
    assert r.status == ref.status == "Solved" and abs(r.iter - ref.iter) <= 25
    assert abs(r.obj_val - ref.obj_val) <= 1e-4 * (1 + abs(ref.obj_val))


@pytest.mark.parametrize("name", sorted(PROBLEMS))
def test_fused_direction_and_product_kernel_is_bit_identical(name, monkeypatch):
    """k_cg_dirA (direction update rebuilt at the gathered columns of the A product, {r, u} interleaved for one 16-byte gather) against
    the separate k_cg_dir + k_spmv_A_rho launches: same iterates, same Krylov iteration counts, bit for bit."""
    prob = PROBLEMS[name]()
    r1, s1, k1, _ = _solve(monkeypatch, "0", prob, 80, fuse="1")
    r0, s0, k0, _ = _solve(monkeypatch, "0", prob, 80, fuse="0")
    assert r1.iter == r0.iter == 80 and r1.kkt_iters_total == r0.kkt_iters_total and k1 == k0
    for a, b in ((r1.x, r0.x), (r1.s, r0.s), (r1.y, r0.y), (s1, s0)):
        assert np.array_equal(a.view(np.int64), b.view(np.int64))
    assert r1.obj_val == r0.obj_val and r1.info.rho_updates == r0.info.rho_updates
