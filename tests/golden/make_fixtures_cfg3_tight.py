"""Generates tests/golden/baseline_cfg3_tight.npz: BASELINE.json config 3 at FULL size -- all 1024 independent SOCPs (n = 500, m = 1000,
50 SecondOrderCone(20) each, seeds 1000..2023) -- run for 60 ADMM iterations with a TIGHT CG (tol_constant = 1e-10, tol_exponent = 0, eps = 0)
through the compiled C restatement of the loop (oracle/cosmo_oracle_c.c: src/solver.jl:137-176, SOC projection src/convexset.jl:100-114,
restated cg!).  With the default 1 / k^1.5 tolerance single problems' trajectories are only comparable to a check interval (the NumPy and the
compiled oracle disagree with each other there); with an exact KKT solve every one of the 1024 trajectories is pinned at 1e-7, which is what the
full-size GPU test compares against (tests/test_gpu_batch.py::test_cfg3_full_size_tight_cg_all_problems).

Full vectors are too large to commit (1024 x 2500 doubles = 20 MB), so per problem the fixture holds: 24 seeded sample entries of x, s and y
(unscaled results), their infinity and 2-norms, the objective, the total Krylov iterations, the rho updates and the 50 SOC branch ids of the LAST
projection.  Like the other fixtures this pins the RESTATED algorithm (the Julia reference cannot run here).
~ 8 min on 5 cores.  Usage:  python tests/golden/make_fixtures_cfg3_tight.py [nproc]"""
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

NPROB, ITERS, NSAMPLE = 1024, 60, 24
SETTINGS = dict(kkt_solver="cg", tol_constant=1e-10, tol_exponent=0.0, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9, max_iter=ITERS)


def sample_idx(size, k, tag):
    return np.sort(np.random.default_rng(7919 * k + tag).choice(size, size=NSAMPLE, replace=False))


def one(k):
    import cosmo_jl_amd as cj            # problem generator only; no device code is touched
    from oracle import cosmo_oracle as O
    from oracle import cosmo_oracle_c as OC
    from tests import util
    p = cj.problems.socp(seed=1000 + k)
    ws = O.Workspace(p["P"], p["q"], p["A"], p["b"], util.oracle_cones(p["sets"]), O.Settings(**SETTINGS))
    c = OC.run(ws)
    row = {}
    for tag, key in enumerate(("x", "s", "y")):
        v = c[key]
        row[key + "_val"] = v[sample_idx(v.size, k, tag)]
        row[key + "_norm"] = np.array([np.max(np.abs(v)), np.linalg.norm(v)])
    row["scalars"] = np.array([c["iter"], c["obj_val"], c["cg_iters_total"], len(c["rho_updates"]), c["r_prim"], c["r_dual"]])
    row["rho_updates"] = np.array((list(c["rho_updates"]) + [0.0] * 4)[:4])
    row["soc_branch"] = np.array([c["soc_branch"][ic] for ic in sorted(c["soc_branch"])], dtype=np.int8)
    assert c["status"] == "Max_iter_reached" and c["iter"] == ITERS, (k, c["status"], c["iter"])
    return k, row


if __name__ == "__main__":
    import subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    t0 = time.time()
    rows = [None] * NPROB
    with Pool(nproc) as pool:
        for i, (k, row) in enumerate(pool.imap_unordered(one, range(NPROB), chunksize=4)):
            rows[k] = row
            if i % 64 == 63:
                print("%d / %d problems, %.0f s" % (i + 1, NPROB, time.time() - t0), flush=True)
    out = {key: np.stack([r[key] for r in rows]) for key in rows[0]}
    out["meta"] = np.array([NPROB, ITERS, NSAMPLE])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "baseline_cfg3_tight.npz"), **out)
    print("done: %d problems x %d iterations, %.0f s, Krylov iterations per problem mean %.0f max %.0f"
          % (NPROB, ITERS, time.time() - t0, out["scalars"][:, 2].mean(), out["scalars"][:, 2].max()), flush=True)
