#!/usr/bin/env python
"""BASELINE config 5 to the default tolerances with the reference's DEFAULT accelerator (AndersonAccelerator, mem 15, safeguarded) on the device:
literal cg! and the opt-in Jacobi-preconditioned CG; the plain loop's record is profiles/r03_cfg5_convergent_device.json (Solved in 2825 iterations, 8.6 s)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cosmo_jl_amd as cj

p = cj.problems.chordal_sdp()
out = {}
for name, kkt in (("anderson_cg", cj.CGIndirectKKTSolver), ("anderson_cg_jacobi", cj.CGJacobiKKTSolver)):
    md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], cj.Settings(kkt_solver=kkt, accelerator=cj.AndersonAccelerator, max_iter=6000, time_limit=60.0))
    t0 = time.time(); cj.model.setup(md); ts = time.time() - t0
    t1 = time.time(); r = cj.optimize(md); tsolve = time.time() - t1
    a = md.handle.accel_stats()
    out[name] = dict(status=r.status, iter=int(r.iter), safeguarding_iter=int(r.safeguarding_iter), obj_val=float(r.obj_val), r_prim=float(r.info.r_prim), r_dual=float(r.info.r_dual),
                     rho_updates=len(r.info.rho_updates) - 1, kkt_iters_total=int(r.kkt_iters_total), setup_seconds=round(ts, 2), solve_seconds=round(tsolve, 2),
                     iter_time=round(r.times.iter_time, 3), accel=a)
    print(name, out[name], flush=True)
os.makedirs("gpurun_out/r04", exist_ok=True)
json.dump(out, open("gpurun_out/r04/cfg5_accelerated_device.json", "w"), indent=1)
