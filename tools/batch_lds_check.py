"""Batch mode: LDS-resident kernel vs the streaming kernel (csrc/batch.hip).  (1) bitwise equality of the iterates at workgroup size
256 (identical arithmetic and accumulation order by construction), (2) agreement at 512 / 1024 threads (different reduction tree),
(3) timing of BASELINE config 3 (1024 SOCPs) for every variant.  Usage: python tools/batch_lds_check.py [nprob] [iters]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np          # noqa: E402
import cosmo_jl_amd as cj   # noqa: E402

nprob = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
probs = [cj.problems.socp(seed=1000 + k) for k in range(nprob)]


def run(env, n_it, count, tight=False):
    for k in ("COSMO_HIP_BATCH_LDS", "COSMO_HIP_BATCH_BS", "COSMO_HIP_BATCH_REG"):
        os.environ.pop(k, None)
    os.environ.update(env)
    kw = dict(kkt_solver=cj.with_options(cj.CGIndirectKKTSolver, tol_constant=1e-10, tol_exponent=0.0)) if tight else {}
    st = cj.Settings(max_iter=n_it, eps_abs=0.0, eps_rel=0.0, **kw)
    mods = []
    for p in probs[:count]:
        md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], st); mods.append(md)
    t = time.time()
    res = cj.optimize_batch(mods)
    return res, time.time() - t


out = {}
ref, _ = run({"COSMO_HIP_BATCH_LDS": "0"}, 80, 64)
ref_t, _ = run({"COSMO_HIP_BATCH_LDS": "0"}, 120, 64, tight=True)
res_t, _ = run({}, 120, 64, tight=True)
out["check_reg_tight_cg"] = dict(max_dx=max(float(np.max(np.abs(a.x - b.x))) for a, b in zip(ref_t, res_t)),
                                 max_ds=max(float(np.max(np.abs(a.s - b.s))) for a, b in zip(ref_t, res_t)),
                                 max_dy=max(float(np.max(np.abs(a.y - b.y))) for a, b in zip(ref_t, res_t)),
                                 same_rho=all(np.allclose(a.info.rho_updates, b.info.rho_updates, rtol=1e-9) for a, b in zip(ref_t, res_t)),
                                 cg_counts_equal=sum(a.kkt_iters_total == b.kkt_iters_total for a, b in zip(ref_t, res_t)), scale_x=float(np.max(np.abs(ref_t[0].x))))
for bs in ("256", "512", "1024"):
    res, _ = run({"COSMO_HIP_BATCH_LDS": "1", "COSMO_HIP_BATCH_BS": bs, "COSMO_HIP_BATCH_REG": "0"}, 80, 64)
    dx = max(float(np.max(np.abs(a.x - b.x))) for a, b in zip(ref, res))
    ds = max(float(np.max(np.abs(a.s - b.s))) for a, b in zip(ref, res))
    same_k = all(a.kkt_iters_total == b.kkt_iters_total for a, b in zip(ref, res))
    bit = all(np.array_equal(a.x, b.x) and np.array_equal(a.s, b.s) and np.array_equal(a.y, b.y) for a, b in zip(ref, res))
    out["check_bs%s" % bs] = dict(max_dx=dx, max_ds=ds, same_cg_counts=same_k, bitwise=bit)
for name, env in (("stream", {"COSMO_HIP_BATCH_LDS": "0"}), ("lds512", {"COSMO_HIP_BATCH_BS": "512", "COSMO_HIP_BATCH_REG": "0"}), ("reg512", {})):
    res, wall = run(env, iters, nprob)
    out[name] = dict(iter_time_s=res[0].times.iter_time, batch_iters_per_s=iters / res[0].times.iter_time,
                     mean_cg=float(np.mean([r.kkt_iters_total / (r.iter + 1) for r in res])), wall_s=round(wall, 2))
print(json.dumps(out, indent=1))
