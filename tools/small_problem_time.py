#!/usr/bin/env python
"""Per-iteration time of ONE small problem: multi-kernel device loop (cj.optimize) vs the persistent one-workgroup kernel of the
batch path (cj.optimize_batch with a batch of one)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch  # noqa
import cosmo_jl_amd as cj
from tests import util
for (n, meq, mineq, mbox) in [(20, 2, 10, 10), (200, 10, 150, 150), (1000, 50, 800, 800), (4000, 100, 3000, 3000)]:
    rng = np.random.default_rng(n)
    prob = util.random_qp(rng, n, meq, mineq, mbox, p_shift=2.0)
    st = dict(max_iter=400, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9)
    res = {}
    for mode in ("loop", "batch1"):
        model = cj.Model(); model.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], cj.Settings(**st))
        t0 = time.perf_counter()
        if mode == "loop":
            r = cj.optimize(model)
        else:
            r = cj.optimize_batch([model])[0]
        res[mode] = (r.times.iter_time if hasattr(r.times, "iter_time") else time.perf_counter() - t0, r.iter, r.kkt_iters_total)
    print("n=%5d m=%5d nnzA=%7d: loop %.1f us/iter (K=%.1f)   batch-of-one %.1f us/iter (K=%.1f)" % (
        n, prob["A"].shape[0], prob["A"].nnz, 1e6 * res["loop"][0] / res["loop"][1], res["loop"][2] / res["loop"][1],
        1e6 * res["batch1"][0] / res["batch1"][1], res["batch1"][2] / res["batch1"][1]), flush=True)
