"""Time to solution with the reference's DEFAULT settings (eps 1e-5, adaptive rho, Ruiz scaling 10, check_termination 25) on the BASELINE
instances at full size, with and without the reference's default accelerator: status, iterations, objective, setup and solve seconds.
The CPU side of the same solves is tests/golden/baseline_convergent.json (oracle_seconds).  usage: gpu_convergent_solves.py [cfg2 cfg4 cfg5]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cosmo_jl_amd as cj

GEN = {"cfg2": (cj.problems.sparse_box_qp, 1000), "cfg4": (cj.problems.closest_correlation, 400), "cfg5": (cj.problems.chordal_sdp, 700)}
out = {}
for name in (sys.argv[1:] or ["cfg2", "cfg4", "cfg5"]):
    gen, mi = GEN[name]
    p = gen()
    for acc in (None, cj.AndersonAccelerator):
        md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], cj.Settings(kkt_solver=cj.CGIndirectKKTSolver, max_iter=mi, accelerator=acc))
        t0 = time.time(); cj.model.setup(md); ts = time.time() - t0
        t1 = time.time(); r = cj.optimize(md); tsolve = time.time() - t1
        key = name + ("+anderson" if acc else "")
        out[key] = dict(status=r.status, iter=int(r.iter), obj_val=float(r.obj_val), r_prim=float(r.info.r_prim), r_dual=float(r.info.r_dual),
                        rho_updates=[float(v) for v in r.info.rho_updates], kkt_iters_total=int(r.kkt_iters_total), x_norm=float(np.linalg.norm(r.x)),
                        setup_seconds=round(ts, 2), solve_seconds=round(tsolve, 2), iter_time=round(r.times.iter_time, 3))
        print(key, out[key], flush=True)
os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "gpu_convergent.json"), "w"), indent=1)
