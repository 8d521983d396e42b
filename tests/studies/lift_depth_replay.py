#!/usr/bin/env python
"""CPU replay study: how many LIFTING steps of the matrix-sign iteration (csrc/psd_polar.hip) would each PSD projection of a
real ADMM run have needed to pass the a-posteriori verification ||(U^2 - I) X||_F <= 2 * 8 d eps ||X||_F?

The device applies the same odd quintic to every eigenvalue, so the answer is a SCALAR computation on the spectrum of every matrix the loop
projects: u0 = 2 g lambda / ||X||_F (g = the spectral rescaling of large cones, 2 / ||U0^2||_F^(1/2)), k lifting steps + the finishing steps from
the table cosmo_hip_polar_schedule exports, then the verified quantity sum lambda^2 (1 - u^2)^2.  The oracle (dsyevr) supplies the spectra.
Output per configuration: histogram of the minimal passing k, and the products per projection of three policies replayed on the sequence --
fixed k = 10 (today), an oracle-minimal k (lower bound), and the two-way adaptive rule the verdict sketches (k - 1 after M verified projections,
k + 1 and a fallback round of 3 + 5 steps on a failure).  Build the adaptive depth only if the replay saves >= 15 % of the products.

usage: lift_depth_replay.py cfg4|cfg5|cfg4small|cfg5small [iterations]
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cosmo_jl_amd as cj            # noqa: E402  (problem generators + the host-side schedule table; no device code runs)
from oracle import cosmo_oracle as O  # noqa: E402
from tests import util               # noqa: E402

EPS = 2.220446049250313e-16
NFIN, RLIFT, KMAX = 5, 3, 16


def schedule(k):
    lib = cj.load_library()
    n = C.c_int32(0)
    abc = (C.c_double * (3 * (k + NFIN)))()
    assert lib.cosmo_hip_polar_schedule(C.c_int32(k), abc, C.byref(n)) == 0 and n.value == k + NFIN
    return np.array(abc[:]).reshape(-1, 3)


LIFT = schedule(1)[0]
FIN = schedule(0)


def apply_steps(u, steps):
    for a, b, c in steps:
        y = u * u
        u = u * (a + b * y + c * y * y)
    return u


def passes(lam, nrm, u, d):
    g2 = float(np.sum(lam * lam * (1.0 - u * u) ** 2))
    return 0.5 * np.sqrt(g2) <= 8.0 * d * EPS * nrm


def minimal_k(lam, large):
    """Smallest number of lifting steps after which the verification passes (large cones: the first step carries the spectral rescaling and from
    d >= 1024 counts double, as polar_enqueue_project does)."""
    d = lam.size
    nrm = float(np.sqrt(np.sum(lam * lam)))
    if nrm == 0.0:
        return 0
    u = 2.0 * lam / nrm
    if large:
        g = 2.0 / np.sqrt(np.sqrt(np.sum((u * u) ** 2)))
        u = u * max(g, 1.0)
    for k in range(0, KMAX + 1):
        if passes(lam, nrm, apply_steps(u.copy(), FIN), d):
            return k
        u = apply_steps(u, [LIFT])
    return KMAX + 1


def products(k, large, d):
    """products of a projection whose main schedule has k lifting steps (+ 2 for the verification)"""
    kk = k
    if large and k > 0 and d >= 1024 and k > 2:
        kk = k - 1                      # the rescaled first step replaces two plain ones from d = 1024 on
    return 3 * (kk + NFIN) + 2


def main(which, iters):
    small = which.endswith("small")
    if which.startswith("cfg4"):
        prob = cj.problems.closest_correlation(d=400 if small else 2000)
    else:
        prob = cj.problems.chordal_sdp(**(dict(ncliques=40, n_total=6000, n_zero=100, n_nonneg=500) if small else {}))
    spectra = []                        # (iteration, cone side, eigenvalues)
    orig = O._psd_project_dense
    it = [0]

    def spy(X):
        w = O._lapack.dsyevr(X, compute_v=0, range="A", lower=0, abstol=-1.0, overwrite_a=0)[0]
        spectra.append((it[0], X.shape[0], w.copy()))
        return orig(X)
    O._psd_project_dense = spy
    st = O.Settings(max_iter=iters, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9, kkt_solver="cg")
    ws = O.Workspace(prob["P"], prob["q"], prob["A"], prob["b"], util.oracle_cones(prob["sets"]), st)
    ncones = sum(1 for c in ws.cones if c.kind in (O.PSD_TRIANGLE, O.PSD_SQUARE) and c.dim > 1)
    t0 = time.time()
    ws.optimize()
    O._psd_project_dense = orig
    # group by projection call: the loop projects every cone once per iteration (+ once in nothing else: the init step has no projection)
    per_it = {}
    for idx, (_, d, w) in enumerate(spectra):
        per_it.setdefault(idx // ncones, []).append((d, w))
    large = not which.startswith("cfg5")
    need = np.array([[minimal_k(w, d > 256) for d, w in cones] for _, cones in sorted(per_it.items())])     # iterations x cones
    dims = np.array([d for d, _ in per_it[0]])
    print("%s: %d iterations x %d PSD cones, oracle %.0f s" % (which, need.shape[0], need.shape[1], time.time() - t0))
    hist = np.bincount(need.ravel(), minlength=KMAX + 2)
    print("minimal passing k, histogram over all (iteration, cone):", {int(k): int(c) for k, c in enumerate(hist) if c})
    kmax_it = need.max(axis=1)          # a batch runs ONE schedule per projection unless the depth is per cone
    print("per iteration max over the cones:", kmax_it.tolist())
    w3 = dims.astype(float) ** 3        # cost weight of a cone's products
    # policies, in products per projection weighted by d^3 (what the product launches cost)
    def cost_fixed(k):
        tot = 0.0
        for row in need:
            fail = row > k
            tot += np.sum(w3 * products(k, large, dims.max())) + np.sum(w3[fail] * (3 * (RLIFT + NFIN) + 2))     # failed cones take one fallback round (they alone: gated tiles)
            fail2 = row > k + RLIFT
            tot += np.sum(w3[fail2] * (3 * (RLIFT + NFIN) + 2))
        return tot / (need.shape[0] * np.sum(w3))
    def cost_oracle_min():
        return float(np.mean([np.sum(w3 * np.array([products(int(k), large, dims.max()) for k in row])) / np.sum(w3) for row in need]))
    def cost_adaptive(M=5, per_cone=True, k0=10):
        k = np.full(need.shape[1], k0) if per_cone else np.array([k0])
        streak = np.zeros_like(k)
        tot = 0.0
        for row in need:
            r = row if per_cone else np.array([row.max()])
            ww = w3 if per_cone else np.array([np.sum(w3)])
            fail = r > k
            tot += np.sum(ww * np.array([products(int(x), large, dims.max()) for x in k]))
            tot += np.sum(ww[fail] * (3 * (RLIFT + NFIN) + 2))
            tot += np.sum(ww[r > k + RLIFT] * (3 * (RLIFT + NFIN) + 2))
            streak = np.where(fail, 0, streak + 1)
            k = np.where(fail, np.minimum(k + 1, KMAX), k)
            down = (streak >= M) & (k > 0)
            k = np.where(down, k - 1, k); streak = np.where(down, 0, streak)
        return tot / (need.shape[0] * np.sum(w3))
    res = dict(config=which, iterations=int(need.shape[0]), cones=int(need.shape[1]),
               histogram={int(k): int(c) for k, c in enumerate(hist) if c},
               products_per_projection=dict(fixed_k10=round(cost_fixed(10), 2), fixed_k9=round(cost_fixed(9), 2), fixed_k8=round(cost_fixed(8), 2),
                                            oracle_minimal_per_cone=round(cost_oracle_min(), 2),
                                            adaptive_per_cone_M5=round(cost_adaptive(5, True), 2), adaptive_per_cone_M20=round(cost_adaptive(20, True), 2),
                                            adaptive_whole_batch_M5=round(cost_adaptive(5, False), 2), adaptive_whole_batch_M20=round(cost_adaptive(20, False), 2)))
    print(json.dumps(res))
    if os.environ.get("LIFT_REPLAY_DUMP"):
        np.savez_compressed(os.environ["LIFT_REPLAY_DUMP"], need=need, dims=dims)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "cfg4small", int(sys.argv[2]) if len(sys.argv) > 2 else 60)
