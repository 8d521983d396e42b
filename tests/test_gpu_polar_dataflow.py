"""The main schedule of the batched sign iteration as ONE persistent, dependency-driven launch (csrc/psd_polar.hip: k_polar_dataflow; round 6, VERDICT r05
item 2) against the launch-per-product form.  Reference semantics: the PsdConeTriangle projections of the composite set (src/convexset.jl:219-263, 885-891)
are independent of each other, so product p + 1 of a cone depends on product p of THAT cone only.  Same tile arithmetic in both forms => the iterates must be
the same BITS; what differs is scheduling (per-XCD in-order queue, per-cone completion counters, operands read with sc1 loads past the L1)."""
import numpy as np
import pytest

import cosmo_jl_amd as cj

pytestmark = pytest.mark.gpu


def _run(prob, iters, monkeypatch, dataflow, dtype=np.float64):
    if dataflow is None:
        monkeypatch.delenv("COSMO_HIP_POLAR_DATAFLOW", raising=False)
    else:
        monkeypatch.setenv("COSMO_HIP_POLAR_DATAFLOW", dataflow)
    st = cj.Settings(max_iter=iters, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9)
    md = cj.Model(dtype=dtype); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], st)
    r = cj.optimize(md)
    h = md.handle
    out = (r, h.polar_stats(), h.polar_dataflow_stats())
    h.close()
    return out


def test_forced_on_a_small_batch_deep_pipelining_is_bit_identical(monkeypatch):
    """40 cliques: ~17 tiles per XCD for 128 persistent workgroups per XCD, so the FIRST item of most workgroups already belongs to a product p > 0 and up to
    seven products of a cone chain are in flight at once -- every hand-off is read microseconds after it was written, by another CU, out of the XCD's L2.
    (With plain operand loads behind `buffer_inv sc0` this case returned NaN: bench/dataflow_lab.hip, profiles/r06_polar_dataflow.txt.)"""
    prob = cj.problems.chordal_sdp(ncliques=40, n_total=6000, n_zero=100, n_nonneg=500)
    r0, p0, d0 = _run(prob, 40, monkeypatch, "0")
    r1, p1, d1 = _run(prob, 40, monkeypatch, "1")
    assert d0["enabled"] == 0 and d0["launches"] == 0
    assert d1["enabled"] == 1 and d1["launches"] >= 40 and d1["products_per_launch"] == p1["products_last_batch"] == p0["products_last_batch"]
    assert r0.iter == r1.iter == 40
    assert np.array_equal(r0.x, r1.x) and np.array_equal(r0.s, r1.s) and np.array_equal(r0.y, r1.y)
    assert p1["unverified"] == 0 and p1["verified"] == p0["verified"]
    # default: a batch this small keeps the launch-per-product form (nothing to overlap, a dependency wait per tile: measured 1.48 -> 1.79 ms per iteration)
    _, _, dd = _run(prob, 2, monkeypatch, None)
    assert dd["enabled"] == 0


def test_default_on_baseline_config_5_is_bit_identical_including_repair_rounds(monkeypatch):
    """BASELINE config 5 (400 cliques, 188 tiles per XCD): the persistent form is the default; 50 iterations include projections whose verification fails and
    is repaired by launch-per-product rounds behind the persistent launch.  Same bits as COSMO_HIP_POLAR_DATAFLOW=0, same verification record."""
    prob = cj.problems.chordal_sdp()
    r1, p1, d1 = _run(prob, 50, monkeypatch, None)
    r0, p0, d0 = _run(prob, 50, monkeypatch, "0")
    assert d1["enabled"] == 1 and d1["launches"] >= 50 and d1["timed_launches"] >= 1 and 0.0 < d1["avg_launch_seconds"] < 0.05
    assert d1["workgroups"] == 1024 and d1["tiles_per_product"] >= 8 * 128
    assert d0["enabled"] == 0
    assert np.array_equal(r0.x, r1.x) and np.array_equal(r0.s, r1.s) and np.array_equal(r0.y, r1.y)
    assert p0["fallback_rounds"] == p1["fallback_rounds"] and p0["verified"] == p1["verified"] and p1["unverified"] == 0
    assert r0.kkt_iters_total == r1.kkt_iters_total


def test_float32_library_persistent_form_is_bit_identical(monkeypatch):
    """libcosmo_hip_f32.so (COSMO.Model{Float32}): the same kernel instantiated for float -- `v_mfma_f32_16x16x4_f32` tiles, 8-byte sc1 buffer loads of the
    operand pairs -- forced on for a 40-clique batch: the same bits as the launch-per-product form of that library."""
    prob = cj.problems.chordal_sdp(ncliques=40, n_total=6000, n_zero=100, n_nonneg=500)
    r0, p0, d0 = _run(prob, 30, monkeypatch, "0", dtype=np.float32)
    r1, p1, d1 = _run(prob, 30, monkeypatch, "1", dtype=np.float32)
    assert d0["enabled"] == 0 and d1["enabled"] == 1 and d1["launches"] >= 30
    assert np.array_equal(r0.x, r1.x) and np.array_equal(r0.s, r1.s) and np.array_equal(r0.y, r1.y)
    assert p1["unverified"] == 0 and p1["verified"] == p0["verified"] and p1["fallback_rounds"] == p0["fallback_rounds"]
