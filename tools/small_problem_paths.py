#!/usr/bin/env python
"""Lab: a SMALL problem (the reference's portfolio example, 210 variables, 211 rows) solved one model at a time through (a) the single-problem handle --
a chain of dependent launches per ADMM iteration -- and (b) the batch path with ONE model (`optimize_batch([model])`: one persistent workgroup, the
problem in LDS, the iterates in registers).  Same settings; per-problem set-up and solve time, time per ADMM iteration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import cosmo_jl_amd as cj

n_assets, k = 200, 10
rng = np.random.default_rng(1)
Dd = rng.uniform(size=n_assets) * np.sqrt(k)
F = sp.random(n_assets, k, density=0.5, random_state=rng, data_rvs=rng.standard_normal).tocsc()
mu = (3.0 + 9.0 * rng.uniform(size=n_assets)) / 100.0
P = sp.block_diag([2.0 * sp.diags(Dd), 2.0 * sp.identity(k)]).tocsc()
A = sp.vstack([sp.hstack([F.T, -sp.identity(k)]), sp.hstack([sp.csr_matrix(np.ones((1, n_assets))), sp.csr_matrix((1, k))]),
               sp.hstack([-sp.identity(n_assets), sp.csr_matrix((n_assets, k))])]).tocsc()
b = np.concatenate([np.zeros(k), [1.0], np.zeros(n_assets)])
sets = [cj.ZeroSet(k + 1), cj.Nonnegatives(n_assets)]
st = cj.Settings(eps_abs=1e-6, eps_rel=1e-6)
gammas = np.logspace(-2, 1, 40)


def model(g):
    md = cj.Model(); md.set(P, np.concatenate([-mu / g, np.zeros(k)]), A, b, sets, st)
    return md


cj.optimize(model(1.0)); cj.optimize_batch([model(1.0)])            # warm both paths
for label, solve in (("single-problem handle (cj.optimize)", lambda md: cj.optimize(md)), ("batch path, one model (cj.optimize_batch([md]))", lambda md: cj.optimize_batch([md])[0])):
    t0 = time.perf_counter(); rs = [solve(model(g)) for g in gammas]; dt = time.perf_counter() - t0
    its = sum(r.iter for r in rs); loop = sum(r.times.iter_time for r in rs)
    print("%-50s %6.1f ms per problem (loop %6.1f ms), %6.1f us per ADMM iteration, %d iterations in total, %d Solved"
          % (label, 1e3 * dt / len(gammas), 1e3 * loop / len(gammas), 1e6 * loop / its, its, sum(r.status == "Solved" for r in rs)), flush=True)
