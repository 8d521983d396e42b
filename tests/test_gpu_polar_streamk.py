"""Opt-in stream-K product kernel (COSMO_HIP_POLAR_STREAMK=1) of the large PSD cones (csrc/psd_polar.hip, k_symm_gemm_sk; reference semantics: project!(::PsdConeTriangle),
/root/reference/src/convexset.jl:402-412 with _project! :219-241).  The (tile, k-panel) work of one symmetric product is cut into equal
contiguous ranges over two resident workgroups per CU; tiles that straddle a range boundary are finished from partial tiles exchanged
through a scratch slot.  Checked here: the projection against LAPACK at sizes that exercise one ticket class (few tiles), eight
classes, ranges with two and with many segments per tile; agreement with the one-tile-per-workgroup kernels (same operands, different
summation grouping: rounding-level differences only); run-to-run bit reproducibility (the ranges are fixed, whichever workgroup
takes them); no spin time-outs."""
import numpy as np
import pytest
import scipy.sparse as sp

import cosmo_jl_amd as cj
from oracle import cosmo_oracle as O
from tests import util

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps


def handle_for(sets, dtype=np.float64):
    m = sum(K.dim for K in sets)
    h = cj.Handle(0, dtype)
    h.set_problem(sp.identity(2, format="csc"), np.zeros(2), sp.csc_matrix((m, 2)), np.zeros(m))
    h.set_cones([K.kind for K in sets], [K.dim for K in sets], None, None)
    return h


def random_sym(rng, d, rank_pos):
    Q = np.linalg.qr(rng.standard_normal((d, d)))[0]
    lam = np.concatenate([rng.uniform(0.1, 2.0, rank_pos), -rng.uniform(0.1, 2.0, d - rank_pos)])
    X = (Q * lam) @ Q.T
    return (X + X.T) / 2


# d -> (tiles of 96, expected ticket classes): 257 -> 6 tiles (1 class), 700 -> 36 tiles (1 class), 1100 -> 78 tiles (8 classes),
# 1500 -> 136 tiles (8 classes)
@pytest.mark.parametrize("d,classes", [(257, 1), (300, 1), (700, 1), (1100, 8), (1500, 8)])
def test_streamk_projection_vs_lapack_and_vs_tile_kernels(d, classes, monkeypatch):
    rng = np.random.default_rng(d)
    K = cj.PsdConeTriangle(d * (d + 1) // 2)
    X = random_sym(rng, d, d // 3)
    s = cj.problems.svec(X)
    ref = s.copy(); info = {}
    O.project(ref, util.oracle_cones([K]), info)
    outs = {}
    for streamk in (1, 0):
        monkeypatch.setenv("COSMO_HIP_POLAR_STREAMK", str(streamk))
        h = handle_for([K])
        st = h.polar_stats(); sk = h.polar_streamk_stats()
        if streamk:
            nt = -(-d // 96)
            assert st["k_split"] == 3 and st["tile_side"] == 96 and sk["enabled"] == 1 and sk["classes"] == classes
            assert sk["workgroups"] % classes == 0 and 1 <= sk["workgroups"] <= 512
            assert sk["workgroups"] // classes <= (nt * (nt + 1) // 2 // classes) * (nt * 96 // 16)      # never more workgroups than units in a class
        else:
            assert st["k_split"] in (1, 2) and sk["enabled"] == 0
        out, rk, _ = h.project(s)
        out2, rk2, _ = h.project(s)
        assert np.array_equal(out.view(np.int64), out2.view(np.int64))            # bit-reproducible, whichever workgroup took which range
        assert np.linalg.norm(out - ref) <= 64 * d * EPS * np.linalg.norm(X)
        assert int(rk[0]) == int(rk2[0]) == info["psd_rank"][0] == d // 3
        assert h.polar_stats()["unverified"] == 0
        if streamk:
            assert h.polar_streamk_stats()["timeouts"] == 0
        outs[streamk] = out
        h.close()
    # same algorithm, same operands; only the grouping of the k-sums differs between the kernels
    assert np.linalg.norm(outs[1] - outs[0]) <= 8 * d * EPS * np.linalg.norm(X)


def test_streamk_several_large_cones_and_fallback_rounds(monkeypatch):
    """Two large cones of different size in one composite set (each gets its own workgroup count; the scratch is shared, the cones run
    one after the other) and a schedule that is too short on purpose (COSMO_HIP_POLAR_KLIFT=0), so that the gated fallback rounds really
    execute through the stream-K kernel."""
    monkeypatch.setenv("COSMO_HIP_POLAR_KLIFT", "0")
    monkeypatch.setenv("COSMO_HIP_POLAR_STREAMK", "1")
    rng = np.random.default_rng(77)
    ds = [420, 900]
    sets = [cj.PsdConeTriangle(d * (d + 1) // 2) for d in ds]
    mats = []
    for d in ds:
        Q = np.linalg.qr(rng.standard_normal((d, d)))[0]
        lam = np.concatenate([rng.uniform(0.5, 2, d // 2), -10.0 ** rng.uniform(-5, -3, d - d // 2)])      # small negative eigenvalues need lifting
        M = (Q * lam) @ Q.T
        mats.append((M + M.T) / 2)
    s = np.concatenate([cj.problems.svec(M) for M in mats])
    ref = s.copy(); O.project(ref, util.oracle_cones(sets))
    h = handle_for(sets)
    out, rk, _ = h.project(s)
    st = h.polar_stats()
    assert st["fallback_rounds"] >= 1 and st["unverified"] == 0
    off = 0
    for K, M, d in zip(sets, mats, ds):
        assert np.linalg.norm(out[off:off + K.dim] - ref[off:off + K.dim]) <= 64 * d * EPS * np.linalg.norm(M)
        off += K.dim
    assert h.polar_streamk_stats()["timeouts"] == 0
    h.close()


def test_streamk_float32_library(monkeypatch):
    monkeypatch.setenv("COSMO_HIP_POLAR_STREAMK", "1")
    d = 700
    rng = np.random.default_rng(5)
    K = cj.PsdConeTriangle(d * (d + 1) // 2)
    X = random_sym(rng, d, 300).astype(np.float32).astype(np.float64)
    s = cj.problems.svec(X)
    ref = s.copy(); info = {}
    O.project(ref, util.oracle_cones([K]), info)
    h = handle_for([K], np.float32)
    assert h.polar_stats()["k_split"] == 3
    out, rk, _ = h.project(s.astype(np.float32))
    e32 = np.finfo(np.float32).eps
    assert np.linalg.norm(out.astype(np.float64) - ref) <= 64 * d * e32 * np.linalg.norm(X)
    assert int(rk[0]) == 300 and h.polar_streamk_stats()["timeouts"] == 0
    h.close()


def test_batched_path_second_tile_class(monkeypatch):
    """Opt-in COSMO_HIP_POLAR_BATCH_TS96=1: cones with sides in (64, 96] and (128, 192] run their products on 96 x 96 tiles (a second
    launch per product), the others keep 64 x 64 tiles; same projection within the 64 d eps bound, exact ranks on gapped spectra."""
    monkeypatch.setenv("COSMO_HIP_POLAR_BATCH_TS96", "1")
    rng = np.random.default_rng(96)
    ds = [70, 96, 100, 130, 192, 200, 33, 64]
    sets = [cj.PsdConeTriangle(d * (d + 1) // 2) for d in ds]
    mats = [random_sym(rng, d, d // 2) for d in ds]
    s = np.concatenate([cj.problems.svec(M) for M in mats])
    ref = s.copy(); info = {}
    O.project(ref, util.oracle_cones(sets), info)
    h = handle_for(sets)
    assert h.polar_stats()["batch_cones"] == len(ds)
    out, rk, _ = h.project(s)
    off = 0
    for K, M, d, r_ref, r in zip(sets, mats, ds, info["psd_rank"], rk):
        assert np.linalg.norm(out[off:off + K.dim] - ref[off:off + K.dim]) <= 64 * d * EPS * np.linalg.norm(M), d
        assert int(r) == r_ref == d // 2
        off += K.dim
    assert h.polar_stats()["unverified"] == 0
    h.close()


def test_wave_per_tile_batched_product_is_bit_identical(monkeypatch):
    """Opt-in COSMO_HIP_POLAR_BATCH_WAVE=1: one wave per 64 x 64 tile, operands loaded in fragment layout straight from global memory (no LDS,
    no barrier in the main loop).  Same instruction, same k order => the projections agree with the workgroup-per-tile kernel bit for bit."""
    rng = np.random.default_rng(3)
    ds = [17, 33, 64, 65, 100, 128, 130, 160, 192, 200, 256]
    sets = [cj.PsdConeTriangle(d * (d + 1) // 2) for d in ds]
    s = np.concatenate([cj.problems.svec(random_sym(rng, d, d // 2)) for d in ds])
    outs = {}
    for w in ("0", "1"):
        monkeypatch.setenv("COSMO_HIP_POLAR_BATCH_WAVE", w)
        h = handle_for(sets)
        out, rk, _ = h.project(s)
        assert h.polar_stats()["unverified"] == 0
        outs[w] = (out, rk)
        h.close()
    assert np.array_equal(outs["0"][0].view(np.int64), outs["1"][0].view(np.int64))
    assert np.array_equal(outs["0"][1], outs["1"][1]) and list(outs["1"][1]) == [d // 2 for d in ds]
