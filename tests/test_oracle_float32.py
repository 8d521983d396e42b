"""CPU: the oracle's cone projections in the element type of the slice -- the reference runs its set tests for every `T in [Float32, Float64,
BigFloat]` (/root/reference/test/run_cosmo_tests.jl:9, test/UnitTests/sets.jl:26-111).  Float32 slices are projected in Float32 (ssyevr, float32
arithmetic), exactly as `COSMO.Model{Float32}` does; the assertions are the reference's membership assertions at its tolerances plus agreement
with the Float64 projection of the same (Float32-rounded) input at a few ulps of Float32."""
import numpy as np
import pytest

from oracle import cosmo_oracle as O


def svec(M):
    d = M.shape[0]
    jj, ii = np.tril_indices(d)
    return np.where(ii == jj, M[ii, jj], np.sqrt(2.0) * M[ii, jj])


@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_projections_land_in_the_sets(T):
    rng = np.random.default_rng(7)
    tol = 1e-4 if T == np.float32 else 1e-9
    # sets.jl:33,39  ZeroSet / Nonnegatives
    x = rng.standard_normal(10).astype(T); O.project_cone(x, O.ZeroSet(10)); assert x.dtype == T and not x.any()
    x = rng.standard_normal(10).astype(T); O.project_cone(x, O.Nonnegatives(10)); assert x.dtype == T and x.min() >= 0
    # sets.jl:50  Box
    l = (-rng.uniform(0, 1, 10)).astype(T); u = rng.uniform(0, 1, 10).astype(T)
    x = (3 * rng.standard_normal(10)).astype(T); O.project_cone(x, O.Box(l, u)); assert x.dtype == T and np.all(x >= l) and np.all(x <= u)
    # sets.jl:59  SecondOrderCone: ||x[2:]|| <= x[1] after the projection (all three branches)
    for scale in (1.0, -5.0, 0.1):
        x = rng.standard_normal(12).astype(T); x[0] = T(scale)
        O.project_cone(x, O.SecondOrderCone(12))
        assert x.dtype == T and np.linalg.norm(x[1:].astype(np.float64)) <= float(x[0]) + tol
    # sets.jl:66-78  PsdCone / PsdConeTriangle: min eig >= -1e-9 (Float64); the Float32 run is held to a Float32 tolerance
    d = 12
    G = rng.standard_normal((d, d)); M = (G + G.T) / 2
    x = M.reshape(-1, order="F").astype(T); O.project_cone(x, O.PsdCone(d * d))
    assert x.dtype == T and np.linalg.eigvalsh(x.reshape(d, d, order="F").astype(np.float64)).min() >= -tol
    xt = svec(M).astype(T); info = {}
    O.project_cone(xt, O.PsdConeTriangle(d * (d + 1) // 2), info)
    X = np.zeros((d, d)); jj, ii = np.tril_indices(d)
    X[ii, jj] = np.where(ii == jj, xt, xt / np.sqrt(2.0)); X = X + np.triu(X, 1).T
    assert xt.dtype == T and np.linalg.eigvalsh(X).min() >= -tol and info["psd_rank"][0] == int((np.linalg.eigvalsh(M) > 0).sum())


def test_float32_projection_agrees_with_float64_on_the_rounded_input():
    rng = np.random.default_rng(8)
    e32 = np.finfo(np.float32).eps
    d = 40
    G = rng.standard_normal((d, d)); M = ((G + G.T) / 2).astype(np.float32)
    x32 = svec(M.astype(np.float64)).astype(np.float32)
    x64 = x32.astype(np.float64)
    K = O.PsdConeTriangle(d * (d + 1) // 2)
    O.project_cone(x32, K); O.project_cone(x64, K)
    assert np.linalg.norm(x32 - x64) <= 64 * d * e32 * np.linalg.norm(x64 if np.linalg.norm(x64) > 0 else 1.0) + 64 * d * e32 * np.linalg.norm(M)
    s32 = rng.standard_normal(20).astype(np.float32); s32[0] = 0.3
    s64 = s32.astype(np.float64)
    O.project_cone(s32, O.SecondOrderCone(20)); O.project_cone(s64, O.SecondOrderCone(20))
    assert np.max(np.abs(s32 - s64)) <= 16 * e32 * np.max(np.abs(s64))
