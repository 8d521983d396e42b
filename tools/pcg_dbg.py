import os, sys
sys.path.insert(0, "/root/repo")
os.environ["COSMO_HIP_CG_PERSIST"] = "1"
import numpy as np
import cosmo_jl_amd as cj
prob = cj.problems.sparse_box_qp(n=3000, m=6000, nnz=50000, seed=9)
st = cj.Settings(max_iter=5, eps_abs=0.0, eps_rel=0.0, check_infeasibility=10 ** 9)
md = cj.Model(); md.set(prob["P"], prob["q"], prob["A"], prob["b"], prob["sets"], st)
cj.model.setup(md)
h = md.handle
print("after setup", h.cg_persist_stats())
rhs = np.random.default_rng(5).standard_normal(md.n + md.m)
sol, its = h.kkt_solve(rhs)
print("after kkt_solve", its, h.cg_persist_stats(), h.get_stats())
