import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cosmo_jl_amd as cj
nprob = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
probs = [cj.problems.socp(seed=1000 + k) for k in range(nprob)]
def run(env, n_it, count):
    for k in ("COSMO_HIP_BATCH_LDS", "COSMO_HIP_BATCH_BS", "COSMO_HIP_BATCH_REG"):
        os.environ.pop(k, None)
    os.environ.update(env)
    st = cj.Settings(max_iter=n_it, eps_abs=0.0, eps_rel=0.0)
    mods = []
    for p in probs[:count]:
        md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], st); mods.append(md)
    res = cj.optimize_batch(mods)
    return res[0].times.iter_time, float(np.mean([r.kkt_iters_total for r in res])), int(np.max([r.kkt_iters_total for r in res]))
for count in (1024,):
    for n_it in (50, 200, 800):
        t, kk, kmax = run({}, n_it, count)
        print("reg  count %4d iters %4d time %.4f s  per-iter %.1f us  mean total CG %.0f max %d" % (count, n_it, t, 1e6 * t / n_it, kk, kmax), flush=True)
