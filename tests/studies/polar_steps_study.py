"""CPU study for the matrix-sign PSD projection (csrc/psd_polar.hip): how many phase-1 steps does the projection need on the matrices
an ADMM run actually feeds it?  Runs the oracle on a closest-correlation problem, captures every matrix handed to the PSD projection,
emulates the device iteration in NumPy (U0 = X / ||X||_F, K1 steps of the (3.4445, -4.7750, 2.0315) quintic, 4 Newton-Schulz steps,
X+ = (X + U X) / 2) and reports max ||dX+||_F / ||X||_F over the trajectory for several K1.  No device code is involved."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cosmo_jl_amd as cj            # noqa: E402  (problem generators)
from oracle import cosmo_oracle as O  # noqa: E402
from tests import util               # noqa: E402


def polar_project(X, k1, k2=4):
    nrm = np.linalg.norm(X)
    if nrm == 0:
        return X.copy()
    U = X / nrm
    for it in range(k1 + k2):
        a, b, c = (3.4445, -4.7750, 2.0315) if it < k1 else (15 / 8, -10 / 8, 3 / 8)
        Y = U @ U
        U = U @ (a * np.eye(len(X)) + b * Y + c * (Y @ Y))
        U = (U + U.T) / 2
    return (X + U @ X) / 2


d = int(sys.argv[1]) if len(sys.argv) > 1 else 60
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 150
if len(sys.argv) > 3 and sys.argv[3] == "chordal":          # a small decomposed SDP: many cliques of side d/2 .. d
    p = cj.problems.chordal_sdp(ncliques=12, dmin=max(4, d // 2), dmax=d, n_total=600, n_zero=20, n_nonneg=60, seed=6)
else:
    p = cj.problems.closest_correlation(d=d, seed=4)
captured = []
orig = O._psd_project_dense


def spy(X):
    captured.append(np.triu(X) + np.triu(X, 1).T)
    return orig(X)


O._psd_project_dense = spy
ws = O.Workspace(p["P"], p["q"], p["A"], p["b"], util.oracle_cones(p["sets"]), O.Settings(max_iter=iters, eps_abs=0, eps_rel=0, check_infeasibility=10 ** 9))
ws.optimize()
O._psd_project_dense = orig
print("captured %d projections (cone sides %d .. %d)" % (len(captured), min(len(X) for X in captured), max(len(X) for X in captured)))
sample = captured[:: max(1, len(captured) // 40)]
lam_rel = []
for X in sample:
    w = np.linalg.eigvalsh(X)
    lam_rel.append(np.sort(np.abs(w))[:3] / np.linalg.norm(X))
lam_rel = np.array(lam_rel)
print("smallest |lambda| / ||X||_F over the sampled iterates: min %.1e median %.1e" % (lam_rel[:, 0].min(), np.median(lam_rel[:, 0])))
for k1 in (20, 18, 16, 14, 12, 10):
    errs = []
    for X in sample:
        w, V = np.linalg.eigh(X)
        ref = (V * np.maximum(w, 0)) @ V.T
        errs.append(np.linalg.norm(polar_project(X, k1) - ref) / np.linalg.norm(X))
    print("K1 = %2d: max rel error %.2e  median %.2e   (test bound 64 d eps = %.1e)" % (k1, max(errs), float(np.median(errs)), 64 * d * 2.2e-16))
