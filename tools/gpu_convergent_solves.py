import sys, time, json
sys.path.insert(0,'/root/repo')
import numpy as np
import cosmo_jl_amd as cj
out={}
for name, gen, mi in (("cfg4", cj.problems.closest_correlation, 400), ("cfg5", cj.problems.chordal_sdp, 700)):
    p = gen()
    md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], cj.Settings(kkt_solver=cj.CGIndirectKKTSolver, max_iter=mi))
    t=time.time(); r = cj.optimize(md)
    out[name]=dict(status=r.status, iter=int(r.iter), obj_val=float(r.obj_val), r_prim=float(r.info.r_prim), r_dual=float(r.info.r_dual), rho_updates=[float(v) for v in r.info.rho_updates], kkt_iters_total=int(r.kkt_iters_total), x_norm=float(np.linalg.norm(r.x)), x_absmax=float(np.max(np.abs(r.x))), seconds=round(time.time()-t,2), polar=md.handle.polar_stats())
    print(name, out[name], flush=True)
json.dump(out, open('/root/repo/gpurun_out/gpu_convergent.json','w'), indent=1)
