"""Generates tests/golden/baseline_cfg4.npz / baseline_cfg5.npz: BASELINE.json configs 4 (closest correlation, d = 2000) and 5
(decomposed chordal SDP, 400 cliques, n = 50 000) at their FULL size, run through the CPU oracle (LAPACK dsyevr projections,
restated cg!) for a few ADMM iterations with tight CG and eps = 0, plus one composite projection of a seeded vector.
The oracle needs ~25 s per cfg5 iteration (m = 2.86 M row vectors in NumPy), which is why the GPU suite compares against these
committed fixtures instead of re-running it on the GPU box; full vectors are too large to commit (cfg4: 2 M doubles), so a
fixture holds 4096 seeded sample entries of every vector, the vector norms and the scalars.  Like oracle_trajectories.npz these
pin the RESTATED algorithm (the Julia reference cannot run here).  Usage:  python tests/golden/make_fixtures_baseline.py [cfg4|cfg5]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cosmo_jl_amd as cj            # noqa: E402  (problem generators only; no device code is touched)
from oracle import cosmo_oracle as O  # noqa: E402
from tests import util               # noqa: E402

SETTINGS = dict(kkt_solver="cg", tol_constant=1e-10, tol_exponent=0.0, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9)
ITERS = {"cfg4": 8, "cfg5": 6}
NSAMPLE = 4096


def problem(name):
    return cj.problems.closest_correlation() if name == "cfg4" else cj.problems.chordal_sdp()


def sample_idx(size, seed):
    return np.sort(np.random.default_rng(seed).choice(size, size=min(NSAMPLE, size), replace=False))


def projection_input(name, m):
    """The vector whose composite projection is pinned (seeded; identical bytes in the generator and in the GPU test)."""
    return np.random.default_rng(77 if name == "cfg4" else 78).standard_normal(m)


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg4", "cfg5"]
    for name in which:
        t0 = time.time()
        p = problem(name)
        ws = O.Workspace(p["P"], p["q"], p["A"], p["b"], util.oracle_cones(p["sets"]), O.Settings(max_iter=ITERS[name], **SETTINGS))
        r = ws.optimize()
        out = {}
        for key, v in (("x", r.x), ("s", r.s), ("y", r.y)):
            idx = sample_idx(v.size, 1 + len(key) + v.size % 97)
            out[key + "_idx"] = idx; out[key + "_val"] = v[idx]; out[key + "_norm"] = np.array([np.linalg.norm(v), np.max(np.abs(v))])
        out["scalars"] = np.array([r.iter, r.r_prim, r.r_dual, r.obj_val, float(np.sum(r.cg_iters))])
        out["rho_updates"] = np.array(r.rho_updates)
        # one composite projection of a seeded vector (src/convexset.jl:885-891 over all cones)
        v = projection_input(name, ws.m)
        info = {}
        O.project(v, ws.cones, info)
        idx = sample_idx(v.size, 5)
        out["proj_idx"] = idx; out["proj_val"] = v[idx]; out["proj_norm"] = np.array([np.linalg.norm(v), np.max(np.abs(v))])
        out["proj_rank"] = np.array(info.get("psd_rank", []), dtype=np.int64)
        # per-cone Frobenius norms of the projected blocks: a checksum that sees every entry
        offs = np.concatenate([[0], np.cumsum([c.dim for c in ws.cones])])
        out["proj_cone_norm"] = np.array([np.linalg.norm(v[offs[k]:offs[k + 1]]) for k in range(len(ws.cones))])
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "baseline_%s.npz" % name), **out)
        print(name, r.status, r.iter, "%.9e" % r.obj_val, "cg its", int(np.sum(r.cg_iters)), "%.0f s" % (time.time() - t0), flush=True)
