"""CPU tests of the matrix-sign schedule of csrc/psd_polar.hip (host function cosmo_hip_polar_schedule, no GPU needed):
the exported table equals what the design tool derives (Remez), it converges on scalars over the designed range, and a NumPy
emulation of the device iteration (same products, same verification bound) reproduces the LAPACK projection."""
import ctypes as C
import os
import sys

import numpy as np

import cosmo_jl_amd as cj

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import polar_schedule as PS  # noqa: E402


def table(k_lift):
    lib = cj._ffi.load_library()
    n = C.c_int32(0)
    assert lib.cosmo_hip_polar_schedule(k_lift, None, C.byref(n)) == 0
    abc = np.zeros(3 * n.value)
    assert lib.cosmo_hip_polar_schedule(k_lift, abc.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n)) == 0
    return abc.reshape(-1, 3)


def test_table_matches_the_design_tool():
    T = table(9)
    assert T.shape == (14, 3)
    a, b, c, e = PS.optimal_quintic(0.022120539994561712, 2.1)        # lifting polynomial: minimax on [l*, 2.1] with p(l*) = 0.085
    assert np.allclose(T[0], [a, b, c], rtol=1e-10) and abs(e - 0.915) < 1e-9
    assert all(np.array_equal(T[t], T[0]) for t in range(9))
    fin = PS.schedule(0.08, 2.02, target=3e-15)
    assert len(fin) == 5
    for t in range(5):
        assert np.allclose(T[9 + t], fin[t][:3], rtol=1e-9), t
    # the lifting polynomial maps [l*, 2.1] into [0.085, 1.915] (no interior root, 5 % margin at the top), slope 3.8438 at 0
    x = np.linspace(0.022120539994561712, 2.1, 400001)
    p = x * (T[0, 0] + x * x * (T[0, 1] + T[0, 2] * x * x))
    assert p.min() >= 0.085 - 1e-9 and p.max() <= 1.915 + 1e-9 and abs(T[0, 0] - 3.8438259784) < 1e-9


def test_scalar_convergence_over_the_designed_range():
    k = 10
    T = table(k)
    lo = 0.085 / (2 * (0.9996 * T[0, 0]) ** k)                         # smallest |lambda| / ||X||_F that k steps lift to 0.085
    assert 5e-8 < lo < 7e-8
    x = 2.0 * np.concatenate([np.geomspace(lo, 1.0, 20001), [1.0]])    # U_0 = 2 X / ||X||_F
    for a, b, c in T:
        x = x * (a + x * x * (b + c * x * x))
        assert x.min() > 0 and x.max() <= 2.0 + 1e-9
    assert np.max(np.abs(x - 1.0)) <= 5e-15
    # below the designed range the eigenvalue is NOT converged -- that is what the verification product detects
    y = np.array([2.0 * lo / 50])
    for a, b, c in T:
        y = y * (a + y * y * (b + c * y * y))
    assert y[0] < 0.9


def emulate(X, k_lift, rounds=2, tol_factor=8.0):
    """The device iteration in NumPy: products, verification bound ||(U^2 - I) X||_F / 2, guarded fallback rounds."""
    d = X.shape[0]
    nf = np.linalg.norm(X)
    U = 2.0 * X / nf
    lift, fin = table(k_lift)[0], table(k_lift)[k_lift:]

    def step(U, co):
        Y = U @ U
        Tm = co[2] * (Y @ Y) + co[1] * Y
        return U @ Tm + co[0] * U
    for _ in range(k_lift):
        U = step(U, lift)
    for co in fin:
        U = step(U, co)
    used = 0
    for r in range(rounds + 1):
        H = U @ X
        err = 0.5 * np.linalg.norm(U @ H - X) / nf
        if err <= tol_factor * d * np.finfo(float).eps or r == rounds:
            break
        used += 1
        for _ in range(3):
            U = step(U, lift)
        for co in fin:
            U = step(U, co)
    return (X + H) / 2, err, used


def test_emulated_projection_matches_lapack_and_fallback_triggers():
    rng = np.random.default_rng(3)
    d = 120
    Q = np.linalg.qr(rng.standard_normal((d, d)))[0]
    eps = np.finfo(float).eps
    # (a) spectrum inside the designed range: no fallback, error at rounding level
    lam = np.concatenate([rng.uniform(0.1, 2, 50), -rng.uniform(0.1, 2, 40), 10.0 ** rng.uniform(-5, -2, 30) * rng.choice([-1, 1], 30)])
    X = (Q * lam) @ Q.T; X = (X + X.T) / 2
    ref = (Q * np.maximum(lam, 0)) @ Q.T
    Xp, err, used = emulate(X, 10)
    assert used == 0 and err <= 8 * d * eps
    assert np.linalg.norm(Xp - ref) <= 64 * d * eps * np.linalg.norm(X)
    assert np.linalg.norm(Xp - ref) <= 2.5 * err * np.linalg.norm(X) + 20 * eps * np.linalg.norm(X)   # the bound is a bound
    # (b) eigenvalues below the range of a SHORT main schedule: the verification fails, one fallback round repairs it
    Xp, err, used = emulate(X, 2)
    assert used >= 1 and err <= 8 * d * eps
    assert np.linalg.norm(Xp - ref) <= 64 * d * eps * np.linalg.norm(X)
    # (c) eigenvalues at the rounding level never lift and never fail the verification (they perturb X+ by less than themselves)
    lam2 = np.concatenate([rng.uniform(0.1, 2, 60), -rng.uniform(0.1, 2, 30), 1e-15 * rng.standard_normal(30)])
    X2 = (Q * lam2) @ Q.T; X2 = (X2 + X2.T) / 2
    Xp, err, used = emulate(X2, 10)
    assert used == 0
    assert np.linalg.norm(Xp - (Q * np.maximum(lam2, 0)) @ Q.T) <= 64 * d * eps * np.linalg.norm(X2)
