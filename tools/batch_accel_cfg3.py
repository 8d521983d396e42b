#!/usr/bin/env python
"""BASELINE config 3 (1024 SOCPs, n = 500, m = 1000) solved to the default tolerances with and without the reference's default accelerator
(AndersonAccelerator, mem 15, safeguarded) in batch mode: wall time of optimize, iteration counts, accelerator counters.
  python tools/batch_accel_cfg3.py [nprob]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cosmo_jl_amd as cj

nprob = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
probs = [cj.problems.socp(seed=1000 + k) for k in range(nprob)]


def run(st, label):
    mods = []
    for p in probs:
        md = cj.Model(); md.set(p["P"], p["q"], p["A"], p["b"], p["sets"], st); mods.append(md)
    B, _ = cj.model.prepare_batch(mods, 0)
    t0 = time.perf_counter(); rs = B.optimize(); dt = time.perf_counter() - t0
    its = np.array([r.iter for r in rs]); sg = np.array([r.safeguarding_iter for r in rs]); kk = np.array([r.kkt_iters_total for r in rs])
    obj = np.array([r.cost for r in rs]); stat = [cj._ffi.STATUS_NAMES[r.status] for r in rs]
    a = B.accel_stats()
    print("%-12s optimize %.3f s | solved %d / %d | ADMM iterations min / median / max / sum = %d / %d / %d / %d (safeguarding %d) | Krylov iterations sum %d, max per problem %d"
          % (label, dt, sum(s == "Solved" for s in stat), nprob, its.min(), np.median(its), its.max(), its.sum(), sg.sum(), kk.sum(), kk.max()), flush=True)
    if a["accelerated"].sum():
        print("             accelerated steps %d, accepted %d, declined %d, memory restarts %d" % (a["accelerated"].sum(), a["accepted"].sum(), a["declined"].sum(), a["restarts"].sum()))
    B.close()
    return obj


o0 = run(cj.Settings(), "plain")
o1 = run(cj.Settings(accelerator=cj.AndersonAccelerator), "anderson")
print("max relative objective difference between the two runs: %.2e" % np.max(np.abs(o0 - o1) / (1 + np.abs(o0))))
tight = cj.with_options(cj.CGIndirectKKTSolver, tol_constant=1e-10, tol_exponent=0.0)
o2 = run(cj.Settings(kkt_solver=tight), "plain/tightCG")
o3 = run(cj.Settings(kkt_solver=tight, accelerator=cj.AndersonAccelerator), "anderson/tightCG")
print("max relative objective difference (tight CG): %.2e" % np.max(np.abs(o2 - o3) / (1 + np.abs(o2))))
