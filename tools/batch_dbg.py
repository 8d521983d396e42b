import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cosmo_jl_amd as cj
probs = [cj.problems.socp(seed=1000 + k) for k in range(64)]
def run(env, n_it):
    for k in ("COSMO_HIP_BATCH_LDS", "COSMO_HIP_BATCH_BS", "COSMO_HIP_BATCH_REG"):
        os.environ.pop(k, None)
    os.environ.update(env)
    st = cj.Settings(max_iter=n_it, eps_abs=0.0, eps_rel=0.0, kkt_solver=cj.with_options(cj.CGIndirectKKTSolver, tol_constant=1e-10, tol_exponent=0.0))
    mods = []
    for p in probs:
        md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], st); mods.append(md)
    return cj.optimize_batch(mods)
for n_it in (30, 41, 79, 81, 120):
    a = run({"COSMO_HIP_BATCH_LDS": "0"}, n_it); b = run({}, n_it)
    dy = [float(np.max(np.abs(x.y - y.y))) for x, y in zip(a, b)]
    k = int(np.argmax(dy))
    print(n_it, "max dy", max(dy), "prob", k, "|y|", float(np.max(np.abs(a[k].y))), "dx", float(np.max(np.abs(a[k].x - b[k].x))), "ds", float(np.max(np.abs(a[k].s - b[k].s))),
          "rho", a[k].info.rho_updates, b[k].info.rho_updates, "row", int(np.argmax(np.abs(a[k].y - b[k].y))), "iters", a[k].iter, b[k].iter, "kkt", a[k].kkt_iters_total, b[k].kkt_iters_total)
