#!/usr/bin/env python
"""Lab: the reference's portfolio example (examples/portfolio_optimisation.jl:25-46, factor model, n = 200 assets, k = 10 factors) for 256 values of the
risk-aversion parameter in ONE optimize_batch call -- with the cooperative long-row passes of the register kernel, with one thread per row
(COSMO_HIP_BATCH_LONG=0), on the streaming kernel, and as 256 sequential single-problem solves."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import cosmo_jl_amd as cj

n_assets, k, nprob = 200, 10, int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(1)
Dd = rng.uniform(size=n_assets) * np.sqrt(k)
F = sp.random(n_assets, k, density=0.5, random_state=rng, data_rvs=rng.standard_normal).tocsc()
mu = (3.0 + 9.0 * rng.uniform(size=n_assets)) / 100.0
P = sp.block_diag([2.0 * sp.diags(Dd), 2.0 * sp.identity(k)]).tocsc()
A = sp.vstack([sp.hstack([F.T, -sp.identity(k)]), sp.hstack([sp.csr_matrix(np.ones((1, n_assets))), sp.csr_matrix((1, k))]),
               sp.hstack([-sp.identity(n_assets), sp.csr_matrix((n_assets, k))])]).tocsc()
b = np.concatenate([np.zeros(k), [1.0], np.zeros(n_assets)])
sets = [cj.ZeroSet(k + 1), cj.Nonnegatives(n_assets)]
probs = [dict(P=P, q=np.concatenate([-mu / g, np.zeros(k)]), A=A, b=b, sets=sets) for g in np.logspace(-2, 1, nprob)]
st = cj.Settings(eps_abs=1e-6, eps_rel=1e-6)


def models():
    out = []
    for p in probs:
        md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], st); out.append(md)
    return out


def batch(label, env):
    for kk, v in env.items():
        os.environ[kk] = v
    cj.optimize_batch(models()[:4])                      # warm the library
    t0 = time.perf_counter(); rs = cj.optimize_batch(models()); dt = time.perf_counter() - t0
    info = cj.model.LAST_BATCH_INFO
    print("%-52s %7.1f ms wall (optimize %.1f ms), %d Solved, iterations %d .. %d" % (label, 1e3 * dt, 1e3 * info["optimize_seconds"], sum(r.status == "Solved" for r in rs),
                                                                                   min(r.iter for r in rs), max(r.iter for r in rs)), flush=True)
    for kk in env:
        os.environ.pop(kk, None)
    return rs


a = batch("optimize_batch, cooperative long-row passes", {})
b_ = batch("optimize_batch, one thread per row", {"COSMO_HIP_BATCH_LONG": "0"})
c = batch("optimize_batch, streaming kernel", {"COSMO_HIP_BATCH_LDS": "0"})
t0 = time.perf_counter(); seq = [cj.optimize(md) for md in models()]; dt = time.perf_counter() - t0
print("%-52s %7.1f ms wall, %d Solved" % ("%d sequential optimize calls (one handle each)" % nprob, 1e3 * dt, sum(r.status == "Solved" for r in seq)))
print("max |obj(batch) - obj(sequential)| / (1 + |obj|) = %.2e" % max(abs(x.obj_val - y.obj_val) / (1 + abs(y.obj_val)) for x, y in zip(a, seq)))
