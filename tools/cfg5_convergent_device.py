import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import cosmo_jl_amd as cj
p = cj.problems.chordal_sdp()
out = {}
for mi in (25000,):
    md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], cj.Settings(kkt_solver=cj.CGIndirectKKTSolver, max_iter=mi))
    t0 = time.time(); cj.model.setup(md); ts = time.time() - t0
    t1 = time.time(); r = cj.optimize(md); tsolve = time.time() - t1
    out["cfg5"] = dict(status=r.status, iter=int(r.iter), obj_val=float(r.obj_val), r_prim=float(r.info.r_prim), r_dual=float(r.info.r_dual), rho_updates=[float(v) for v in r.info.rho_updates],
                       kkt_iters_total=int(r.kkt_iters_total), x_norm=float(np.linalg.norm(r.x)), setup_seconds=round(ts, 2), solve_seconds=round(tsolve, 2), iter_time=round(r.times.iter_time, 3))
    print(out, flush=True)
json.dump(out, open("gpurun_out/r03/cfg5_convergent_device.json", "w"), indent=1)
