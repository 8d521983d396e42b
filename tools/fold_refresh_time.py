import time, numpy as np, sys
sys.path.insert(0, '/root/repo')
import cosmo_jl_amd as cj
prob = cj.problems.chordal_sdp()
st = cj.Settings(kkt_solver=cj.CGIndirectKKTSolver, max_iter=3, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10**9)
md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], st)
r = cj.optimize(md)
h = md.handle
print(h.fold_stats())
rho = h.get_rho_vec()
for i in range(5):
    t0 = time.perf_counter(); h.update_rho(rho); _ = h.get_rho_classes(); t1 = time.perf_counter()
    print("update_rho + sync: %.3f ms" % ((t1 - t0) * 1e3))
